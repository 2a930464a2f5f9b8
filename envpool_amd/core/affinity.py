"""Host placement on a multi-socket box.

The reference scales across sockets by starting one process per NUMA node under `numactl` (`benchmark/numa_test.sh:15-21`)
and offers `thread_affinity_offset` for its worker threads (`core/async_envpool.h:99-114`).  Here the work is on the GPU,
and what the host side feels is WHICH socket the calling thread runs on: the device's doorbells, its completion signals
and this runtime's pinned memory all live on the device's own NUMA node (HalfCheetah N = 65536, numpy step: 0.50 ms from
the far socket, 0.47 ms from the device's own, `profiles/r6x_numa_probe.txt`).  The library pins only its own helper
threads (engine key "numa_bind"); a process that wants its calling threads there too says so with
`bind_host_to_device()` -- one process per GPU, as `bench.py --gpus N` runs them.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional


def _parse_cpulist(text: str) -> list[int]:
    cpus: list[int] = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.extend(range(int(lo), int(hi or lo) + 1))
    return cpus


def _pci_bus_id(device: int) -> Optional[str]:
    try:
        hip = ctypes.CDLL("libamdhip64.so")
    except OSError:
        return None
    buf = ctypes.create_string_buffer(64)
    if hip.hipDeviceGetPCIBusId(buf, 64, int(device)) != 0:
        return None
    return buf.value.decode().lower()


def device_numa_node(device: int = 0) -> Optional[int]:
    """NUMA node of HIP device `device` (None: no device, single-node host, or sysfs says -1)."""
    bus = _pci_bus_id(device)
    if bus is None:
        return None
    try:
        with open(f"/sys/bus/pci/devices/{bus}/numa_node") as f:
            node = int(f.read())
    except (OSError, ValueError):
        return None
    return node if node >= 0 else None


def device_local_cpus(device: int = 0) -> list[int]:
    """CPUs of the device's NUMA node that this process may use ([] if unknown)."""
    node = device_numa_node(device)
    if node is None:
        return []
    try:
        with open(f"/sys/devices/system/node/node{node}/cpulist") as f:
            cpus = _parse_cpulist(f.read())
    except (OSError, ValueError):
        return []
    allowed = os.sched_getaffinity(0)
    return [c for c in cpus if c in allowed]


def bind_host_to_device(device: int = 0) -> dict:
    """Restrict the calling process (every thread it starts from now on, and the calling thread) to the CPUs of the
    device's NUMA node.  Returns what was done: {"node": n or None, "cpus": count, "bound": bool}."""
    cpus = device_local_cpus(device)
    if not cpus:
        return {"node": device_numa_node(device), "cpus": 0, "bound": False}
    os.sched_setaffinity(0, cpus)
    return {"node": device_numa_node(device), "cpus": len(cpus), "bound": True}

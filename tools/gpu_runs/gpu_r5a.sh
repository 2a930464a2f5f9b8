#!/bin/bash
# Round 5, call a: rocprofv3 evidence for the HBM-streaming kernels (K1 / K2 / K4) at N = 65536 and 4 M, and the
# host-link probe behind the numpy-API figure.
set -u
export TMPDIR=/tmp
O=gpurun_out/r5a
mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
bash tools/profile_families.sh r5a_families --families CartPole,MountainCar,Pendulum,Acrobot,FrozenLake,NChain,Blackjack,Catch > $O/families.log 2>&1
tail -30 $O/families.log
timeout 300 python tools/pcie_probe.py > $O/pcie_probe.jsonl 2> $O/pcie_probe.err; cat $O/pcie_probe.jsonl; tail -3 $O/pcie_probe.err

#!/bin/bash
# Probe a box for the third-party libraries whose arithmetic the MuJoCo / Atari
# oracles restate (MuJoCo 3.6.0, OpenCV 4.13, ALE).  Output is committed under
# profiles/ as evidence of (un)reachability; if a wheel IS found the pinning
# scripts (tools/pin_with_mujoco.py, tools/pin_with_opencv.py) can run there.
echo "== host: $(hostname) $(date -u +%FT%TZ)"
echo "== python imports"
for m in mujoco cv2 ale_py gymnasium dm_control mujoco_py; do
  python - <<P 2>&1
try:
    import $m
    print("$m", "FOUND", getattr($m, "__version__", "?"))
except Exception as e:
    print("$m", "absent:", type(e).__name__, e)
P
done
echo "== shared libraries"
find / -xdev \( -name 'libmujoco*' -o -name 'libopencv_imgproc*' -o -name 'libale*' -o -name 'mujoco*.whl' -o -name 'opencv*.whl' \) -not -path '/proc/*' 2>/dev/null | head -20
echo "(end of find)"
echo "== pip"
timeout 30 pip download mujoco==3.6.0 --no-deps -d /tmp/whl 2>&1 | tail -3
timeout 30 pip index versions mujoco 2>&1 | tail -2
timeout 30 pip download opencv-python-headless --no-deps -d /tmp/whl 2>&1 | tail -2
echo "== network"
timeout 10 python - <<'P' 2>&1
import socket
for h in ("pypi.org", "files.pythonhosted.org", "github.com"):
    try:
        socket.create_connection((h, 443), timeout=3).close(); print(h, "reachable")
    except Exception as e:
        print(h, "unreachable:", e)
P

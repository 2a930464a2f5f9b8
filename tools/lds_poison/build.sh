#!/bin/bash
set -e
cd "$(dirname "$0")"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -fPIC -shared poison.hip -o libpoison.so

cd $GRAFT_REPO_ROOT
export OMP_NUM_THREADS=4
timeout 900 python -m pytest tests/test_gpu_mujoco.py tests/test_gpu_api.py -q -x 2>&1 | tail -3
unset OMP_NUM_THREADS
for t in Walker2d Hopper Ant; do
for p in fp64 fp32; do
timeout 300 python bench.py --task $t --precision $p --no-cpu-baseline 2>&1 | tail -1 | cut -c1-130
done; done
timeout 300 python bench.py --task Ant --num-envs 32768 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-130

#!/bin/bash
# round 4, call d: (1) the MFMA-vs-quad probe of the Humanoid dual-matrix product, with PMC evidence;
# (2) A/B of the split download of the numpy API (EPA_D2H_PARTS = 1 / 2 / 3 / 4)
mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_rows_probe.hip -o gpurun_out/mfma_rows_probe
( gpurun_out/mfma_rows_probe 65536 20; gpurun_out/mfma_rows_probe 16384 50 ) > gpurun_out/r4d_mfma_probe.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_F64 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/r4d_mfma_pmc -o pmc --output-format csv -- $GRAFT_REPO_ROOT/gpurun_out/mfma_rows_probe 65536 3 > $GRAFT_REPO_ROOT/gpurun_out/r4d_mfma_pmc.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY' >> gpurun_out/r4d_mfma_probe.txt 2>&1
import csv, glob, collections
for f in glob.glob("gpurun_out/r4d_mfma_pmc/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]; acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
    for k in acc:
        print("PMC", k, {c: v / max(1, n[(k, c)]) for c, v in acc[k].items()}, "(mean per launch)")
PY
for P in 1 2 3 4; do
  echo "EPA_D2H_PARTS=$P" >> gpurun_out/r4d_d2h_split_ab.txt
  EPA_D2H_PARTS=$P timeout 300 python tools/bench_numpy_api.py HalfCheetah-v4 65536 150 >> gpurun_out/r4d_d2h_split_ab.txt 2>&1
  EPA_D2H_PARTS=$P timeout 300 python tools/bench_numpy_api.py Humanoid-v4 16384 40 >> gpurun_out/r4d_d2h_split_ab.txt 2>&1
done
cat gpurun_out/r4d_mfma_probe.txt gpurun_out/r4d_d2h_split_ab.txt; tail -5 gpurun_out/r4d_mfma_pmc.log

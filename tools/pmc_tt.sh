#!/bin/bash
# usage: tools/pmc_tt.sh <outdir> -- PMC passes over the one-pass retrieval kernel (tools/bench_tt.py)
out=${1:-gpurun_out/pmc_tt}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $out
i=0
while read -r set; do
  i=$((i+1))
  REPS=2 timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $out/p$i -- python tools/bench_tt.py > $out/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $out/p$i.log; }
done <<SETS
FETCH_SIZE
WRITE_SIZE
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS
SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD
SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_LDS
GRBM_GUI_ACTIVE
TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
SETS
python - "$out" <<'PY'
import csv, glob, os, sys, collections, json
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, 'p*', '**', '*counter_collection.csv'), recursive=True):
    for row in csv.DictReader(open(f)):
        name = row['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0]
        acc[name][row['Counter_Name']].append(float(row['Counter_Value']))
res = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}
json.dump(res, open(os.path.join(out, 'pmc_summary.json'), 'w'), indent=1, sort_keys=True)
for k in sorted(res):
    if 'transit' in k or 'interp_ec_batch' in k:
        print(k)
        for c in sorted(res[k]):
            print(f'   {c:36s} {res[k][c]:.4g}')
PY

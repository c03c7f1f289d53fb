#!/bin/bash
# PMC passes over one DiT forward at the bench batch; one rocprofv3 run per counter set (slot limits:
# FETCH_SIZE = 3 TCC slots, WRITE_SIZE = 2).  usage (on the GPU box): pmc_run.sh <tag> [B] [script under tools/dbg]
tag=$1; B=${2:-4096}; script=${3:-dit_fwd.py}
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
out=$root/gpurun_out/pmc_$tag
mkdir -p $out
run() { name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $out/$name -o run -- python3 $root/tools/dbg/$script $B 2 > $out/$name.log 2>&1 || { echo "pass $name failed"; tail -5 $out/$name.log; exit 1; }
  python3 $root/tools/dbg/pmc_agg.py $out/$name $out/$name.json > $out/$name.txt
  find $out/$name -name '*.csv' -size +2M -delete
}
run fetch FETCH_SIZE TCC_HIT_sum && run write WRITE_SIZE TCC_MISS_sum TCC_REQ_sum && \
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE
echo pmc done

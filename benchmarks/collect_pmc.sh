#!/bin/bash
# Collects rocprofv3 counter passes (separate runs, --pmc only, program directly
# after `--`) for the headline bench and for the secondary kernels:
#   bash benchmarks/collect_pmc.sh [bench|secondary|all]
# Output: gpurun_out/pmc/<pass>/...counter_collection.csv ; reduce with
#   python benchmarks/summarise_pmc.py gpurun_out/pmc rNN
# A pass that times out stops the script (no further GPU step after a kill).
export TMPDIR=/tmp
WHAT=${1:-all}
P=gpurun_out/pmc
mkdir -p $P
rocprofv3 -L > $P/counters_available.txt 2>&1 || true

run_pass() {   # name, counters, program...
    local name=$1 ctrs=$2
    shift 2
    timeout -k 10 420 rocprofv3 --pmc $ctrs --output-format csv -d $P/$name -- "$@" > $P/$name.out 2> $P/$name.log
    local rc=$?
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then
        echo "pass $name timed out: stopping" | tee -a $P/status.txt
        exit 1
    fi
    echo "pass $name rc=$rc" | tee -a $P/status.txt
    find $P/$name -name '*kernel_trace.csv' -delete 2>/dev/null
    return 0
}

SQ1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
SQ2="SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM"
SQ3="SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE"

if [ "$WHAT" = bench ] || [ "$WHAT" = all ]; then
    B="python3 bench.py --steps 5 --warmup 3 --no-cpu --no-steady"
    run_pass bench_sq1 "$SQ1" $B
    run_pass bench_sq2 "$SQ2" $B
    run_pass bench_sq3 "$SQ3" $B
    run_pass bench_fetch "FETCH_SIZE" $B
    run_pass bench_write "WRITE_SIZE" $B
fi
if [ "$WHAT" = secondary ] || [ "$WHAT" = all ]; then
    S="python3 benchmarks/bench_kernels.py --pmc-subset"
    run_pass sec_sq1 "$SQ1" $S
    run_pass sec_sq2 "$SQ2" $S
    run_pass sec_sq3 "$SQ3" $S
    run_pass sec_fetch "FETCH_SIZE" $S
    run_pass sec_write "WRITE_SIZE" $S
fi
ls $P

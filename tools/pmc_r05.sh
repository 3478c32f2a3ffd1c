#!/bin/bash
# Round 5: request-path counters of the two sparse passes.  One rocprofv3 --pmc pass per
# counter group (no trace domains mixed in), C3, 3 timed steps.  usage: tools/pmc_r05.sh <outdir> [bench args]
# then: python3 tools/pmc_r05_summary.py <outdir> > profiles/r05_pmc_sparse_passes.txt
# (no TA_* group: with TA_TA_BUSY_sum / TA_ADDR_STALLED_BY_TC_CYCLES_sum / TA_DATA_STALLED_BY_TC_CYCLES_sum /
#  TA_TOTAL_WAVEFRONTS_sum rocprofv3 aborted on this box -- signal 6 in its tool library -- and the pass
#  then sat silent until the run was killed: gpurun_out/r5d/pmc/p4.err, round 5)
# SKIP_PASSES="1 2 3": passes already collected
export TMPDIR=/tmp
out=$1; shift
mkdir -p $out
i=0
while read -r ctrs; do
  [ -z "$ctrs" ] && continue
  i=$((i+1))
  case " $SKIP_PASSES " in *" $i "*) continue;; esac
  echo "pass $i: $ctrs"
  timeout -k 5 240 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out/p$i -- python3 bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 "$@" > $out/p$i.json 2> $out/p$i.err
  rc=$?
  if [ $rc -ge 124 ]; then echo "pass $i timed out or was killed (rc $rc): no further pass"; exit 1; fi
  [ $rc -eq 0 ] || { echo "pass $i failed"; tail -3 $out/p$i.err; }
done <<'LIST'
TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum
TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TOTAL_READ_sum
TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum
TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_TAG_STALL_sum TCC_BUSY_sum
TCC_CYCLE_sum TCC_EA0_RDREQ_LEVEL_sum TCC_SRC_FIFO_FULL_sum TCC_LATENCY_FIFO_FULL_sum
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAVES
GRBM_GUI_ACTIVE GRBM_COUNT
LIST

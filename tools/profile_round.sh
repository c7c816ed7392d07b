#!/bin/bash
# Collects the rocprofv3 evidence bench.py's roofline refers to.  Run ON THE GPU BOX from the repo root:
#   bash tools/profile_round.sh r06
# kernel-trace/stats and every --pmc group are separate runs (gpurun refuses --pmc combined with trace domains).
# The program after `--` is python3 itself (no env / bash -c hop: the profiler's preload initialises the GPU first).
set -e
R=${1:-r06}
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/prof_$R
cd $GRAFT_REPO_ROOT
if [ "$2" = "kt" ]; then rm -rf $O/kt; else rm -rf $O; fi
mkdir -p $O
# which box, which clocks: every pass below runs on THIS box inside this one gpurun call
{ echo "host $(hostname)"; date -u +%FT%TZ; rocm-smi --showclocks 2>/dev/null | grep -E "sclk|mclk|fclk" ; rocm-smi --showproductname 2>/dev/null | grep -E "Card (Series|SKU)|GFX"; } > $O/box.txt 2>&1 || true
# the kernel trace is of the DRIVER's command (python bench.py --gpus 1 --steps 20 --warmup 5), extra legs and cpu baseline included
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/kt.log 2>&1
echo "kernel trace done" > $O/progress.txt
if [ "$2" = "kt" ]; then exit 0; fi
# (counter passes serialise the kernels: the bench's own deadline for the extra legs is lifted, a watchdog exit would lose the profiler's output)
B="python3 bench.py --gpus 1 --steps 3 --warmup 1 --no-extra --no-cpu-baseline --extra-budget-s 3000 --hard-limit-s 3000"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $B > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- $B > $O/write.log 2>&1
echo "traffic done" >> $O/progress.txt
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/sq -- $B > $O/sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/sq2 -- $B > $O/sq2.log 2>&1 || true
# config 3's kernel (k_hash_cells over the 8 GiB slot): HBM bytes against the algorithmic 8 GiB + leaves
H="python3 bench.py --gpus 1 --steps 1 --warmup 0 --no-cpu-baseline --no-child-legs --legs slot_root --extra-budget-s 3000 --hard-limit-s 3000"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/hfetch -- $H > $O/hfetch.log 2>&1 || true
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/hwrite -- $H > $O/hwrite.log 2>&1 || true
# ... and its issue side: the same `sq` group as for k_permute_batch, over the five launches that cover the whole slot
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/hsq -- $H > $O/hsq.log 2>&1 || true
rocprofv3 --pmc SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $O/hsq2 -- $H > $O/hsq2.log 2>&1 || true
echo "hash kernel passes done" >> $O/progress.txt
# saturated issue cost of every opcode of the kernel's stream, with the same counters (tools/ubench_classes.hip; its binary is built in-tree
# by `hipcc --offload-arch=gfx950 -O3 -o tools/ubench_classes tools/ubench_classes.hip` and travels with the snapshot)
if [ -x tools/ubench_classes ]; then
  rocprofv3 --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d $O/ubench -- tools/ubench_classes > $O/ubench.log 2>&1 || true
fi
echo "all done" >> $O/progress.txt
echo "now run locally: python3 tools/ubench_classes_summarize.py $R prof_$R/ubench && python3 tools/valu_roof.py $R && python3 tools/profile_summarize.py $R   (gpurun merges only gpurun_out/ back)"

#!/bin/bash
# Run on the GPU box (through gpurun): the one-rank RCCL worker of tests/test_gpu_rccl.py under rocprofv3 --kernel-trace:
# which kernels did librccl launch on gfx950?  -> gpurun_out/TAG_rccl_kernels.txt
TAG=${1:-r}
REPO=$PWD
export TMPDIR=/tmp
OUT=$REPO/gpurun_out
mkdir -p $OUT
python3 -c "import sys; sys.path.insert(0, 'tests'); import test_gpu_rccl as t; open('/tmp/rccl_worker.py', 'w').write(t.WORKER)"
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29571 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 HSA_ENABLE_IPC_MODE_LEGACY=0 NCCL_DEBUG=VERSION
cd /tmp
timeout -k 5 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_rccl_$TAG -o stats -- python3 /tmp/rccl_worker.py $REPO > $OUT/rccl_$TAG.log 2>&1
cd $REPO
DB=$(find $OUT/prof_rccl_$TAG -name "*.db" | head -1)
python3 tools/rocpd_summary.py $DB $OUT/${TAG}_rccl_all_kernels.txt "tests/test_gpu_rccl.py worker, one rank, nccl backend (rocprofv3 --kernel-trace --stats)" > /dev/null
(echo "# kernels of librccl in the one-rank worker of tests/test_gpu_rccl.py (rocprofv3 --kernel-trace --stats; MI355X, gfx950)"; grep -i "rccl\|nccl\|ncclDevKernel\|AllReduce\|ReduceScatter\|AllGather\|Broadcast" $OUT/${TAG}_rccl_all_kernels.txt | head -40; echo "# worker output:"; grep -v "amdgpu.ids" $OUT/rccl_$TAG.log | tail -8) > $OUT/${TAG}_rccl_kernels.txt
cat $OUT/${TAG}_rccl_kernels.txt | cut -c1-200
rm -rf $OUT/prof_rccl_$TAG

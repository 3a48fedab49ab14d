python - <<'PY'
src=open('tests/test_gpu_multiview.py').read()
w=src.split("SHARDED_WORKER = r'''")[1].split("'''")[0]
open('/tmp/sw.py','w').write(w)
PY
for i in 1 2 3 4; do
MASTER_ADDR=127.0.0.1 OMP_NUM_THREADS=2 python -m torch.distributed.run --nnodes=1 --nproc-per-node=8 --master-addr 127.0.0.1 --master-port $((29700+i)) /tmp/sw.py $PWD 1 8 > gpurun_out/w8_$i.log 2>&1
echo "run $i rc=$?"; grep -v "Gloo\|socket\|^W1\|^\*\*\*" gpurun_out/w8_$i.log | grep -B2 -A6 "Error\b\|assert" | head -40
done

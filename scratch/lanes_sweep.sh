python -m pytest tests/test_gpu_engine.py -x -q 2>&1 | tail -3
for m in 0,0,0,1,1 0,1,2,3,4 0,1,1,2,3 0,1,1,2,2 0,1,2,3,3 0,0,1,2,3; do
  echo "LANES $m"; RTP_LANES=$m python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c1-200
done
echo "HWQ8"; GPU_MAX_HW_QUEUES=8 RTP_LANES=0,1,2,3,4 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c1-200
echo "graph"; RTP_LANES=0,1,2,3,4 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --graph 2>&1 | tail -1 | cut -c1-200

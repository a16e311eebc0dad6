cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c2; O=gpurun_out/c2
timeout 900 python tools/four_models.py fp16 fp16 fp16 fp16 fp16 tiny fp16 fp16x2m fp16x2m fp16x2m > $O/four_models.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "device_feed or stand_in or staged_step" > $O/feed_tests.txt 2>&1
timeout 1500 python tools/rccl_rehearsal.py fp16x2m fp16 > $O/rccl_rehearsal.txt 2>&1
THREADS=512 CS=0,8,16,32 timeout 900 python tools/rccl_rehearsal.py fp16x2m > $O/rccl_rehearsal_512.txt 2>&1
timeout 1200 python bench.py --no-eval-metric --steps 50 --repeats 3 > $O/bench.json 2> $O/bench.err
GPU_MAX_HW_QUEUES=8 timeout 600 python bench.py --no-eval-metric --no-cpu-baseline --steps 50 --repeats 3 > $O/bench_hwq8.json 2> $O/bench_hwq8.err

for i in 1 2 3; do
  python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-times 2>/dev/null | python -c "import sys,json; print('A', json.loads(sys.stdin.readline())['ms_per_step'])" >> gpurun_out/ab.log
  GAEXT_LIB=$PWD/imagenet-models_amd/csrc/libgaext_b.so python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-times 2>/dev/null | python -c "import sys,json; print('B', json.loads(sys.stdin.readline())['ms_per_step'])" >> gpurun_out/ab.log
done

# usage: ab_env.sh VAR VALUE_A VALUE_B   -- same-box A/B of an environment switch, 3 alternating runs of bench.py
for i in 1 2 3; do
  for v in "$2" "$3"; do
    env $1=$v python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-times 2>/dev/null | python -c "import sys,json; print('$1=$v', json.loads(sys.stdin.readline())['ms_per_step'])" >> gpurun_out/ab.log
  done
done

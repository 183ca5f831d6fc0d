for i in 1 2 3; do
  for m in 0 1; do
    echo -n "RU_WZ32=$m step: "; RU_WZ32=$m python3 bench.py --steps 20 --warmup 5 --no-extras 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"
  done
done
for B in 1 4 8; do
for i in 1 2; do
  RU_WZ32=0 python3 tools/fwd_probe.py $B bf16x3 40 2>/dev/null | tail -1 | sed "s/^/[RU_WZ32=0 b$B] /"
  RU_WZ32=1 python3 tools/fwd_probe.py $B bf16x3 40 2>/dev/null | tail -1 | sed "s/^/[RU_WZ32=1 b$B] /"
done
done

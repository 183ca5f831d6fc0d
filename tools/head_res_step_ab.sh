for i in 1 2 3; do
  for m in 0 1; do
    echo -n "RU_HEAD_RES=$m step: "; RU_HEAD_RES=$m python3 bench.py --steps 20 --warmup 5 --no-extras 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"
  done
done

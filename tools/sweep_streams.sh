run() { # label, env..., args
  label=$1; shift
  out=$(env "$@" timeout 120 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-host-frames $EXTRA 2>&1 | tail -1)
  echo "$label $(echo "$out" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d.get('single_stream',{}).get('value'))" 2>&1 | tail -1)"
}
EXTRA="--streams 2"
run base X=1
run tailv7 NUNIF_TAIL_VARIANT=7
run tailv2 NUNIF_TAIL_VARIANT=2
run tailv3 NUNIF_TAIL_VARIANT=3
run attw8 NUNIF_ATTN_WAVES=8
run attw12 NUNIF_ATTN_WAVES=12
run nosnake NUNIF_SNAKE=0
EXTRA="--streams 2 --batch-size 23"
run b23 X=1
EXTRA="--streams 2 --batch-size 15"
run b15 X=1
EXTRA="--streams 4"
run s4 X=1
EXTRA="--streams 3 --batch-size 23"
run s3b23 X=1

run() { # name streams scratch env...
  name=$1; s=$2; g=$3; shift 3
  env GPU_MAX_HW_QUEUES=40 "$@" timeout 150 python bench.py --streams $s --scratch-gb $g --steps 32 --warmup 2 --cpu-seconds 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); st=d['stage_ms_per_step']; print('$name', round(d['value']/1e9,3), round(d['ms_per_step'],1), 'mut', round(st['mutate']), 'fin', round(st['final']), 'chunks', d['roofline']['launches_per_step'], 'miss', d['traceback_window_misses_per_step'], 'passes', d['mutate_passes_per_step'])"
}
run base 8 30
run seg16 8 30 BRX_SEG_WAVES_PER_CU=16
run lane1500 8 30 BRX_LANE_THRESHOLD=1500
run lane8000 8 30 BRX_LANE_THRESHOLD=8000
run s9x26 9 26

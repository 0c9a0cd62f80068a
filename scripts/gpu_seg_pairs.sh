for cfg in "0 0" "150 160" "160 160" "150 128" "144 160"; do set -- $cfg
 OAVIF_AMD_SEG_ROWS=$1 OAVIF_AMD_SEG_ROWS_TAIL=$2 python bench.py --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('seg $1 tail $2', d['value'], d['ms_per_step'], d['stages_ms'], 'one-stream', d['score_roofline']['ms_per_score_device'], 'cached', d['cached_reference']['ms_per_score'])"
done

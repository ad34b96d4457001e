# C2 headline over the plan geometry (slice width x parts ~ 256 workgroups): whole-step and kernel time; on the GPU box
for cfg in "0 0" "15625 4" "11765 3" "7813 2" "3907 1"; do
  set -- $cfg
  python bench.py --steps 200 --warmup 50 --no-cpu --no-secondary ${1:+--width $1} ${2:+--parts $2} 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print(d['config']['plan_slices'], '| value', d['value'], 'ms/step', d['ms_per_step'], 'kernel', d['roofline']['kernel_ms'], d['parity_check']['ok'])"
done

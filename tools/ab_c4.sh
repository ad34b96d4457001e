# usage: tools/ab_c4.sh "<flags A>" "<flags B>" ...  — C4 (binned route, hetero and homo: value, ms/step, dominant-kernel ms) per hipcc
# flag set of be_csr_binned.hip; on the GPU box
set -e
export AB_FILE=brainevent_amd/csrc/be_csr_binned.hip
run() { for h in "" "--homo"; do python bench.py --workload fcn $h --steps 40 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print('C4', '$h', d['value'], d['ms_per_step'], d['roofline'].get('dominant_kernel_ms'))"; done; }
export -f run
bash tools/ab_build.sh "$@" -- bash -c run

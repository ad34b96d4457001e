#!/usr/bin/env python3
"""One scale of examples/coba_2005.py / cuba_2005.py as a replayed HIP graph: us per step (wrap in tools/prof_any.sh for the kernels).
usage: python tools/time_network.py [coba|cuba] [scale] [steps] [unroll]"""
import importlib.util, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
kind = sys.argv[1] if len(sys.argv) > 1 else 'coba'
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20000
unroll = int(sys.argv[4]) if len(sys.argv) > 4 else 10
spec = importlib.util.spec_from_file_location('net', os.path.join(R, 'examples', f'{kind}_2005.py'))
m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
n, el, rate, _, _ = m.run_fused(scale, steps, graph=True, unroll=unroll)
print(f'{kind} scale={scale:g} n={n}: {el / steps * 1e6:.2f} us/step, {rate:.2f} Hz', flush=True)

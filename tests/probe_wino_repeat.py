"""Developer probe (GPU box): the wino unit cases over and over in ONE process -- looks for timing-dependent failures."""
import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import test_gpu_conv as T

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
bad = {}
for r in range(reps):
    for case in T.WINO_CASES:
        try:
            T.test_conv3x3_c64_wino(*case)
        except AssertionError as e:
            bad.setdefault(("plain", case), []).append(str(e)[:120])
    for case in [(256, 12, 20, 1, 1), (128, 8, 32, 2, 0), (256, 18, 26, 2, 1), (256, 2, 2, 1, 1), (256, 136, 240, 2, 1), (256, 24, 336, 13, 1)]:
        try:
            T.test_conv3x3_c64_wino_up2(*case)
        except AssertionError as e:
            bad.setdefault(("up2", case), []).append(str(e)[:120])
print(f"{reps} repetitions; failing cases: {len(bad)}")
for k, v in bad.items():
    print(k, len(v), v[0])

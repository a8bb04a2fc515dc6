import time, sys
sys.path.insert(0, '.')
import torch, elastic_elgamal_amd as eg
pk = bytes.fromhex("a6adb6e9c0ae8d54c26e6e56b5ccd7a16bb0e1951abe4d7ee7028e3d4eca8531")
t0=time.time(); c = eg.Context(0); t1=time.time()
print(f"context (20-bit G table): {1e3*(t1-t0):.1f} ms")
for i in range(3):
    t0=time.time(); p = eg.ChoiceParams(c, pk, 5, True); t1=time.time()
    print(f"params (20-bit K table): {1e3*(t1-t0):.1f} ms")
t0=time.time(); bad = c.selfcheck_generator_table(True, 0); t1=time.time()
print(f"wide G table build: {1e3*(t1-t0):.1f} ms")

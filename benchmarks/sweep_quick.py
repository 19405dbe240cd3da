import sys, json; sys.path.insert(0, '.'); sys.path.insert(0, 'benchmarks')
import sweep_chain as sc
for C in [int(a) for a in sys.argv[1:]] or (4, 8, 16, 32, 64, 128, 256):
    dt, _ = sc.chain(C, steps=40, warm=10)
    print(json.dumps({"channels": C, "ms": dt * 1e3}), flush=True)

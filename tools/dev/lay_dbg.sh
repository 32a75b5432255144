DEBUG_FIRST=1 SEED=23 N=150 python tools/dev/fuzz_parity.py 2>&1 | tail -4

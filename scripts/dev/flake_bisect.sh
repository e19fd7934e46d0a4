run() { f=0; for i in $(seq 1 20); do env "$@" python -m pytest tests/test_harness_pin_gpu.py -m gpu -q --tb=line -p no:cacheprovider -k runner 2>&1 | grep -q "failed" && f=$((f+1)); done; echo "$* : $f of 20 failed"; }
run FOO=1
run FSGS_OVERLAP_VIEWS=0
run FSGS_CACHE_COLORS=0
run FSGS_OVERLAP_VIEWS=0 FSGS_CACHE_COLORS=0
run FSGS_LIB_PATH=$PWD/free-surgs_amd/fsgs_amd/lib/libfsgs_hip.old.so FSGS_CACHE_COLORS=0

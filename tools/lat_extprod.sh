for b in 1 2 4 5; do for w in 256 1; do echo "batch=$b min_wgs=$w"; PFHE_FUSED_MIN_WGS=$w BATCH=$b COEFF_ONLY=1 python tools/perf_extprod.py 2>&1 | grep ext-prod; done; done

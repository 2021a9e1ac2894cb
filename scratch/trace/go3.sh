for b in 512 768 1024; do echo "== WGRAD_BLOCKS=$b"; DYNMM_WGRAD_BLOCKS=$b python scratch/trace/run_trace_wgrad.py 2>&1 | grep wgrad; done

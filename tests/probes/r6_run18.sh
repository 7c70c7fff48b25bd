set -x
for ov in 3 2 0 3 2; do echo "overlap $ov"; PFMI_DEBUG_HOOKS=1 PFMI_DEVCB_OVERLAP=$ov XW_AB_C5=1 timeout 600 bash tests/probes/xw_ab.sh default; done
( timeout 900 python -m pytest tests/test_gpu_elbo.py -q -m gpu -x -k "callback or closure" ) 2>&1 | tail -3

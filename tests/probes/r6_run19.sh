set -x
for lean in 1 0 1 0; do echo "PFX_LEAN=$lean two streams"; PFX_LEAN=$lean XW_AB_C5=1 timeout 600 bash tests/probes/xw_ab.sh default; echo "PFX_LEAN=$lean one stream"; PFX_LEAN=$lean PFMI_DEBUG_HOOKS=1 PFMI_DEVCB_OVERLAP=0 timeout 600 bash tests/probes/xw_ab.sh default; done
( PFX_LEAN=1 timeout 900 python -m pytest tests/test_gpu_elbo.py -q -m gpu -x -k "callback or closure" ) 2>&1 | tail -3

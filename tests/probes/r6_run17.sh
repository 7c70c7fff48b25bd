set -x
for mb in 256 512 1024 2048 4096; do echo "chunk $mb"; PFMI_DEBUG_HOOKS=1 PFMI_DEVCB_CHUNK_MB=$mb timeout 600 bash tests/probes/xw_ab.sh default; done

export PFMI_DEBUG_HOOKS=1
for r in 0 32 48 64 80 96; do echo "reader CUs $r"; PFMI_DEVCB_READER_CUS=$r timeout 300 python tests/probes/devcb_probe.py 8 1000 2>&1 | tail -2; done
for mb in 512 1024 4096; do echo "reader CUs 64 chunk $mb MB"; PFMI_DEVCB_CHUNK_MB=$mb PFMI_DEVCB_READER_CUS=64 timeout 300 python tests/probes/devcb_probe.py 8 1000 2>&1 | tail -2; done

export PFMI_DEBUG_HOOKS=1 PFMI_STREAM_TRACE=1
timeout 300 python tests/probes/api_timeline.py 8 2>&1 | grep "stream:\|wall" | tail -4
timeout 300 python tests/probes/stream_probe.py 8 1000 5 2>&1 | grep "stream:\|streamed" | tail -3

timeout 300 python tests/probes/api_timeline.py 8 2>&1 | tail -22
timeout 300 python tests/probes/api_timeline.py 64 2>&1 | tail -22

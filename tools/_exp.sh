set -e
export PYTHONPATH=.
timeout -k 10 300 python3 -u -m pytest tests/test_parity_gpu.py -x -q --timeout 120 --timeout-method thread 2>&1 | tail -2
for i in 1 2; do timeout -k 10 120 python3 tools/sweep.py config3 26,12 | grep -v "slot rays" | cut -c60-170; done
RT_PARTS=2 timeout -k 10 120 python3 tools/sweep.py config3 26,12 | grep -v "slot rays" | cut -c60-170
RT_PARTS=4 timeout -k 10 120 python3 tools/sweep.py config3 26,12 | grep -v "slot rays" | cut -c60-170
RT_PARTS=8 timeout -k 10 120 python3 tools/sweep.py config3 26,12 | grep -v "slot rays" | cut -c60-170
timeout -k 10 120 python3 tools/sweep.py config5 26,12 | grep -v "slot rays" | cut -c60-170
timeout -k 10 120 python3 tools/sweep.py config2 24,12 | grep -v "slot rays" | cut -c60-170
RT_OVERLAP=0 timeout -k 10 120 python3 tools/sweep.py config3 26,12 | grep -v "slot rays" | cut -c60-170

echo "--- graph"; timeout 200 python -X faulthandler bench.py --no-cpu-baseline --no-kernel-roofline --precondition 0 --steps 5 --warmup 2 2>&1 | grep -v "^  File \"/usr" | tail -40

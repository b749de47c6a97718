# LeRF frame time, alternating library/python variants is not possible in one tree: prints the frame time three times
for i in 1 2 3; do timeout -k 10 300 python tools/scratch/lerf_time.py 2>/dev/null | grep "s/frame" | cut -c1-200; done

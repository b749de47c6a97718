#!/bin/bash
# the C++ drop-in's LeRF training bench repeated under switches: is its loss after the timed steps the same from run to run?
run() { timeout -k 10 200 oracle/_ref/adapter_check bench train_lerf 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], d['loss_first_last'])"; }
for i in 1 2 3; do NRF_MLP_HOST_REPACK=1 run host_repack; done
for i in 1 2 3; do NRF_GEMM_NARROW_ROWS=0 run narrow0; done
for i in 1 2; do run default; done

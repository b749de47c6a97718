#!/usr/bin/env bash
# The LibTorch drop-in on the host's clock (oracle/_ref/adapter_check bench ...): one JSON line per run into gpurun_out/<tag>_dropin.jsonl
#   tools/gpu_dropin_bench.sh <tag> [what ...]     default: all six
set -uo pipefail
tag="${1:-dropin}"; shift || true
what=("$@"); [ ${#what[@]} -eq 0 ] && what=(frame_hash frame_classic frame_lerf train_hash train_classic train_lerf)
mkdir -p gpurun_out
out="gpurun_out/${tag}_dropin.jsonl"; : > "$out"
for w in "${what[@]}"; do
  echo "== $w" >&2
  case "$w" in
    train_*_hipadam) timeout -k 10 300 oracle/_ref/adapter_check bench "${w%_hipadam}" 0 0 hipadam >> "$out" 2> "gpurun_out/${tag}_${w}.err" || echo "{\"bench\": \"$w\", \"rc\": $?}" >> "$out" ;;
    *) timeout -k 10 300 oracle/_ref/adapter_check bench "$w" >> "$out" 2> "gpurun_out/${tag}_${w}.err" || echo "{\"bench\": \"$w\", \"rc\": $?}" >> "$out" ;;
  esac
  tail -n 1 "$out" | cut -c1-1500 >&2
done

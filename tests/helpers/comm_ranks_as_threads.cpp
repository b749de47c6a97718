// comm_ranks_as_threads.cpp -- TEST INFRASTRUCTURE: nrf_comm_* / nrf_allgather_tiles (include/nerfpp_hip.h, comm.hip) at world sizes 2..6 on ONE GPU, the ranks being
// threads of this process over tests/helpers/mock_rccl.cpp (linked in under RCCL's SONAME, so comm.hip's dlopen finds it instead of the real library, which refuses two
// ranks on one device).  Per case: every rank fills its row tile of `frames` images with a pattern of (frame, row, column, channel), all ranks gather, every rank's
// frames must equal the pattern everywhere -- equal tiles (one ncclAllGather per frame), unequal ones (grouped ncclBroadcasts), ranks that own no rows (h < world),
// two gathers back to back on a stream with the tile rewritten in between; then nrf_allreduce_grads at world 2..8: bucketed in-place mean of two gradient buffers against
// host-formed values, twice in a row, and the overflow agreement (one rank reports, every rank skips, nothing is exchanged).  Prints one line per case and "all ok" / "FAILED"; exit code 0 iff all pass.
#include "nerfpp_hip.h"

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#define HIPOK(x) do { if ((x) != hipSuccess) { fprintf(stderr, "HIP call failed: %s (%s:%d)\n", #x, __FILE__, __LINE__); return false; } } while (0)

static float pattern(int gen, int f, int row, int col, int ch) { return (float)(gen * 7 + f) * 1000.0f + (float)row + (float)col * 1e-3f + (float)ch * 1e-5f; }

static bool rank_body(int world, int rank, const char *id, int frames, int h, int w, int ch, std::atomic<int> &errors)
{
    HIPOK(hipSetDevice(0));
    nrf_comm *c = nullptr;
    if (nrf_comm_create_timeout(id, world, rank, 30.0, &c) != NRF_OK) { fprintf(stderr, "rank %d: %s\n", rank, nrf_last_error()); errors++; return false; }
    if (nrf_comm_world(c) != world || nrf_comm_rank(c) != rank) { errors++; return false; }
    int row0 = 0, rows = 0;
    nrf_tile_partition(h, world, rank, &row0, &rows);
    hipStream_t st; HIPOK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const size_t px = (size_t)w * ch, tile_n = (size_t)frames * rows * px, frame_n = (size_t)frames * h * px;
    float *d_tile = nullptr, *d_frames = nullptr;
    if (tile_n) HIPOK(hipMalloc(reinterpret_cast<void **>(&d_tile), tile_n * 4));
    HIPOK(hipMalloc(reinterpret_cast<void **>(&d_frames), (frame_n ? frame_n : 1) * 4));
    std::vector<float> host_tile(tile_n), host_frames(frame_n);
    bool ok = true;
    for (int gen = 0; gen < 2 && ok; gen++) {                     // two gathers back to back: the second's tile upload is ordered behind the first gather on the stream
        for (int f = 0; f < frames; f++) for (int r = 0; r < rows; r++) for (int x = 0; x < w; x++) for (int k = 0; k < ch; k++)
            host_tile[(((size_t)f * rows + r) * w + x) * ch + k] = pattern(gen, f, row0 + r, x, k);
        if (tile_n) HIPOK(hipMemcpyAsync(d_tile, host_tile.data(), tile_n * 4, hipMemcpyHostToDevice, st));
        HIPOK(hipMemsetAsync(d_frames, 0xff, (frame_n ? frame_n : 1) * 4, st));
        if (nrf_allgather_tiles(c, d_tile, frames, h, w, ch, d_frames, st) != NRF_OK) { fprintf(stderr, "rank %d: %s\n", rank, nrf_last_error()); ok = false; break; }
        if (frame_n) HIPOK(hipMemcpyAsync(host_frames.data(), d_frames, frame_n * 4, hipMemcpyDeviceToHost, st));
        HIPOK(hipStreamSynchronize(st));
        for (int f = 0; f < frames && ok; f++) for (int r = 0; r < h && ok; r++) for (int x = 0; x < w && ok; x++) for (int k = 0; k < ch; k++)
            if (host_frames[(((size_t)f * h + r) * w + x) * ch + k] != pattern(gen, f, r, x, k)) { fprintf(stderr, "rank %d: frame %d row %d col %d ch %d wrong (gather %d)\n", rank, f, r, x, k, gen); ok = false; break; }
    }
    if (!ok) errors++;
    if (d_tile) (void)hipFree(d_tile);
    (void)hipFree(d_frames);
    (void)hipStreamDestroy(st);
    nrf_comm_destroy(c);
    return ok;
}

// ---- nrf_allreduce_grads: two gradient buffers (a "table" and a "blob") become their mean over the ranks in place; expected values formed on the host in the mock's own
// order (sum over ranks 0, 1, ... then x 1 / world: for world 2 that is (a + b) * 0.5f -- what the gloo GradSync of nerfpp_amd/dist.py computes, bit for bit).
// `overflow_rank` >= 0: that rank reports an fp16 overflow -> every rank must come back with skip = 1 and its gradients untouched.
static float gpat(int rank, int which, int64_t i) { return (float)((rank + 1) * (which ? 3 : 1)) * 0.25f + (float)(i % 1021) * 1e-3f - (float)((i * 7 + rank) % 13) * 0.0625f; }

static bool allreduce_rank_body(int world, int rank, const char *id, int64_t n0, int64_t n1, int64_t bucket_bytes, int overflow_rank, std::atomic<int> &errors)
{
    HIPOK(hipSetDevice(0));
    nrf_comm *c = nullptr;
    if (nrf_comm_create_timeout(id, world, rank, 30.0, &c) != NRF_OK) { fprintf(stderr, "rank %d: %s\n", rank, nrf_last_error()); errors++; return false; }
    hipStream_t st; HIPOK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const int64_t counts[2] = {n0, n1};
    float *d[2] = {nullptr, nullptr};
    std::vector<float> h[2];
    for (int w = 0; w < 2; w++) {
        h[w].resize((size_t)counts[w]);
        for (int64_t i = 0; i < counts[w]; i++) h[w][(size_t)i] = gpat(rank, w, i);
        if (counts[w]) { HIPOK(hipMalloc(reinterpret_cast<void **>(&d[w]), (size_t)counts[w] * 4)); HIPOK(hipMemcpyAsync(d[w], h[w].data(), (size_t)counts[w] * 4, hipMemcpyHostToDevice, st)); }
    }
    bool ok = true;
    for (int round = 0; round < 2 && ok; round++) {                 // round 1 reduces the already-averaged buffers again (all ranks hold the same values: the mean is themselves)
        int skip = -1;
        const int flag = overflow_rank < 0 ? (round == 0 ? -1 : 0) : (rank == overflow_rank ? 1 : 0);          // round 0 without the agreement (fully asynchronous), round 1 with
        if (nrf_allreduce_grads(c, d, counts, 2, bucket_bytes, flag, &skip, st) != NRF_OK) { fprintf(stderr, "rank %d: %s\n", rank, nrf_last_error()); ok = false; break; }
        std::vector<float> got[2];
        for (int w = 0; w < 2; w++) { got[w].resize((size_t)counts[w]); if (counts[w]) HIPOK(hipMemcpyAsync(got[w].data(), d[w], (size_t)counts[w] * 4, hipMemcpyDeviceToHost, st)); }
        HIPOK(hipStreamSynchronize(st));
        if (overflow_rank >= 0) {
            if (skip != 1) { fprintf(stderr, "rank %d: skip %d, want 1\n", rank, skip); ok = false; }
            for (int w = 0; w < 2 && ok; w++) for (int64_t i = 0; i < counts[w]; i++) if (got[w][(size_t)i] != h[w][(size_t)i]) { fprintf(stderr, "rank %d: gradient touched although the step is skipped\n", rank); ok = false; break; }
            continue;
        }
        if (skip != 0) { fprintf(stderr, "rank %d: skip %d, want 0\n", rank, skip); ok = false; }
        for (int w = 0; w < 2 && ok; w++)
            for (int64_t i = 0; i < counts[w]; i++) {
                float want;
                if (round == 0) { want = gpat(0, w, i); for (int q = 1; q < world; q++) want = want + gpat(q, w, i); want = want * (1.0f / (float)world); h[w][(size_t)i] = want; }
                else { want = h[w][(size_t)i]; float a = want; for (int q = 1; q < world; q++) a = a + want; want = a * (1.0f / (float)world); h[w][(size_t)i] = want; }
                if (got[w][(size_t)i] != want) { fprintf(stderr, "rank %d: buffer %d element %lld: %.9g, want %.9g (round %d)\n", rank, w, (long long)i, got[w][(size_t)i], want, round); ok = false; break; }
            }
    }
    if (!ok) errors++;
    for (int w = 0; w < 2; w++) if (d[w]) (void)hipFree(d[w]);
    (void)hipStreamDestroy(st);
    nrf_comm_destroy(c);
    return ok;
}

int main()
{
    struct Case { int world, frames, h, w, ch; };
    const Case cases[] = {{2, 1, 8, 5, 3}, {2, 2, 9, 4, 3}, {3, 1, 10, 7, 3}, {4, 2, 800, 16, 3}, {4, 1, 801, 8, 3}, {5, 3, 7, 3, 1}, {6, 1, 4, 5, 3}, {3, 2, 2, 6, 3}, {2, 1, 1, 9, 3}};
    int bad = 0;
    for (const Case &cs : cases) {
        char id[NRF_COMM_ID_BYTES];
        if (nrf_comm_unique_id(id) != NRF_OK) { fprintf(stderr, "%s\n", nrf_last_error()); return 2; }
        std::atomic<int> errors{0};
        std::vector<std::thread> th;
        for (int r = 0; r < cs.world; r++) th.emplace_back([&, r] { rank_body(cs.world, r, id, cs.frames, cs.h, cs.w, cs.ch, errors); });
        for (auto &t : th) t.join();
        printf("world %d frames %d h %d w %d ch %d (%s tiles%s): %s\n", cs.world, cs.frames, cs.h, cs.w, cs.ch, cs.h % cs.world ? "unequal" : "equal", cs.h < cs.world ? ", some ranks own no rows" : "",
               errors.load() ? "FAIL" : "ok");
        fflush(stdout);
        bad += errors.load() != 0;
    }
    struct RCase { int world; long long n0, n1, bucket; int overflow_rank; };
    const RCase rcases[] = {{2, 1000003, 17, 1 << 20, -1}, {2, 4096, 1, 0, -1}, {3, 300001, 70000, 1 << 18, -1}, {4, 1 << 20, 18000, 1 << 20, -1}, {6, 12345, 0, 4096, -1}, {8, 100000, 5, 1 << 16, -1},
                            {2, 5000, 7, 0, 1}, {5, 70001, 33, 1 << 16, 3}};
    for (const RCase &rc : rcases) {
        char id[NRF_COMM_ID_BYTES];
        if (nrf_comm_unique_id(id) != NRF_OK) { fprintf(stderr, "%s\n", nrf_last_error()); return 2; }
        std::atomic<int> errors{0};
        std::vector<std::thread> th;
        for (int r = 0; r < rc.world; r++) th.emplace_back([&, r] { allreduce_rank_body(rc.world, r, id, rc.n0, rc.n1, rc.bucket, rc.overflow_rank, errors); });
        for (auto &t : th) t.join();
        printf("allreduce_grads world %d counts %lld + %lld bucket %lld B%s: %s\n", rc.world, rc.n0, rc.n1, rc.bucket, rc.overflow_rank >= 0 ? " (one rank reports overflow: all skip)" : "",
               errors.load() ? "FAIL" : "ok");
        fflush(stdout);
        bad += errors.load() != 0;
    }
    printf("%s %d\n", bad ? "FAILED" : "all ok", bad);
    return bad ? 1 : 0;
}

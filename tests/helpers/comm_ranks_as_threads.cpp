// comm_ranks_as_threads.cpp -- TEST INFRASTRUCTURE: nrf_comm_* / nrf_allgather_tiles (include/nerfpp_hip.h, comm.hip) at world sizes 2..6 on ONE GPU, the ranks being
// threads of this process over tests/helpers/mock_rccl.cpp (linked in under RCCL's SONAME, so comm.hip's dlopen finds it instead of the real library, which refuses two
// ranks on one device).  Per case: every rank fills its row tile of `frames` images with a pattern of (frame, row, column, channel), all ranks gather, every rank's
// frames must equal the pattern everywhere -- equal tiles (one ncclAllGather per frame), unequal ones (grouped ncclBroadcasts), ranks that own no rows (h < world),
// two gathers back to back on a stream with the tile rewritten in between.  Prints one line per case and "all ok" / "FAILED"; exit code 0 iff all pass.
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
    printf("%s %d\n", bad ? "FAILED" : "all ok", bad);
    return bad ? 1 : 0;
}

// scratch.hip -- the library's own stream-ordered scratch memory (round 6).
//
// Short-lived device buffers inside an entry point (the split image of a layer product's B operand, the slice sums of a weight-gradient product, a per-call reduction
// cell) used to come from hipMallocAsync / hipFreeAsync on the caller's stream.  Inside the LibTorch C++ host that was not safe: a training backward issued from the
// autograd engine's thread computed whole layer products from a B image that had been handed out again while the product still read it (adapter_check bench
// train_classic: gradients a few percent off, then NaN; the same calls from the Python mirror were right; a buffer that is never given back to the driver made the
// drop-in right as well) -- the driver pool's reuse rules across threads and streams are not ours to rely on.  Here a block belongs to ONE (device, stream) pair for
// its whole life: whoever takes it next enqueues behind whoever gave it back, on that same stream.  Blocks are kept until scratch_trim() (or process exit); their
// total is the peak of what one stream had in flight at once (tens of MB for the training steps).
#include "common.h"

#include <mutex>
#include <vector>

namespace nrf {

namespace {
struct Block { void *p; size_t size; int device; hipStream_t st; bool busy; };
std::mutex g_mu;
std::vector<Block> g_blocks;
}  // namespace

hipError_t scratch_take(void **out, size_t bytes, hipStream_t st)
{
    if (!out) return hipErrorInvalidValue;
    *out = nullptr;
    if (bytes == 0) bytes = 1;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lk(g_mu);
    Block *best = nullptr;
    for (auto &b : g_blocks)
        if (!b.busy && b.device == dev && b.st == st && b.size >= bytes && (!best || b.size < best->size)) best = &b;
    if (best && best->size <= 4 * bytes + ((size_t)1 << 20)) { best->busy = true; *out = best->p; return hipSuccess; }          // (not a 1 GB block for 16 bytes)
    const size_t size = (bytes + ((size_t)1 << 16) - 1) & ~(((size_t)1 << 16) - 1);
    void *p = nullptr;
    e = hipMalloc(&p, size);
    if (e != hipSuccess) {
        // out of memory: give the idle blocks of this device back and try once more.  (The device is drained as a whole: a block's stream may have been destroyed since
        // -- its handle must not be touched again)
        (void)hipDeviceSynchronize();
        for (size_t i = 0; i < g_blocks.size();) {
            if (!g_blocks[i].busy && g_blocks[i].device == dev) { (void)hipFree(g_blocks[i].p); g_blocks.erase(g_blocks.begin() + (long)i); }
            else i++;
        }
        (void)hipGetLastError();
        e = hipMalloc(&p, size);
        if (e != hipSuccess) return e;
    }
    g_blocks.push_back(Block{p, size, dev, st, true});
    *out = p;
    return hipSuccess;
}

hipError_t scratch_give(void *p, hipStream_t st)
{
    if (!p) return hipSuccess;
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto &b : g_blocks)
        if (b.p == p) { b.busy = false; b.st = st; return hipSuccess; }
    return hipErrorInvalidValue;
}

size_t scratch_trim()
{
    std::lock_guard<std::mutex> lk(g_mu);
    size_t freed = 0;
    int cur = 0; (void)hipGetDevice(&cur);
    int synced = -1;
    for (size_t i = 0; i < g_blocks.size();) {
        if (!g_blocks[i].busy) {
            (void)hipSetDevice(g_blocks[i].device);
            if (synced != g_blocks[i].device) { (void)hipDeviceSynchronize(); synced = g_blocks[i].device; }          // (never the block's stream handle: it may be gone)
            (void)hipFree(g_blocks[i].p);
            freed += g_blocks[i].size;
            g_blocks.erase(g_blocks.begin() + (long)i);
        } else i++;
    }
    (void)hipSetDevice(cur);
    return freed;
}

}  // namespace nrf

/* Gives the library's idle scratch blocks back to the driver (drains the device first); returns the bytes freed.  Optional: the blocks are reused
 * from call to call and amount to the peak of one stream's short-lived buffers. */
extern "C" NRF_API size_t nrf_scratch_trim(void) { return nrf::scratch_trim(); }

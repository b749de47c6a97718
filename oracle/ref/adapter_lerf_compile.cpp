// adapter_lerf_compile.cpp -- TEST INFRASTRUCTURE, compile-only (g++ -c, never linked).
//
// nrfpp::HipLeRFRenderer : LeRFRenderer (include/nerfpp_torch.h) against the reference's LeRFRenderer.h where it lies: every `override` must match a
// virtual of the reference class (RunLENetwork, RawToLEOutputs, RenderRays), the base constructor must accept a null CuHashEmbedder, and the result
// struct must have the members copied into it -- all checked by the compiler.  It cannot be LINKED here: the base class's vtable and its
// Render / BatchifyRays / RenderRays bodies are in LeRFRenderer.cpp, which includes RuCLIPProcessor.h (LeRFRenderer.cpp:2) and calls Relevancy (:79)
// from the external DeliriumV01D/RuCLIP module (absent, no pinned version).  What the subclass forwards to -- nrfpp::HipLeRFPass -- is linked and run
// by adapter_check.
#define NRFPP_WITH_REFERENCE
#define NRFPP_WITH_LERF_RENDERER
#include "nerfpp_torch.h"

// instantiate the members so that their bodies are compiled, not just parsed
nrfpp::HipLeRFRenderer *make_hip_lerf_renderer(nrfpp::HipHashEmbedder e, LeRF lerf) { return new nrfpp::HipLeRFRenderer(e, lerf); }
LeRFRenderResult render_rays_through_it(nrfpp::HipLeRFRenderer &r, torch::Tensor rays) { return r.RenderRays(rays, torch::Tensor(), 64, false, false, 0.f, 128); }

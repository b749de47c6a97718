// adapter_bench.cpp -- TEST INFRASTRUCTURE (built only where /root/reference exists; linked into oracle/_ref/adapter_check).
//
//   adapter_check bench frame_hash    [poses] [dump_dir]     HipNeRFRenderer<HipHashEmbedder, HipSHEncoder, NeRFSmall>::Render(800, 800, K, params, c2w), 64 + 128
//   adapter_check bench frame_classic [poses]                HipNeRFRenderer<HipEmbedder, HipEmbedder, NeRF 8x256>::Render
//   adapter_check bench frame_lerf    [poses]                HipLeRFPass::Render (what HipLeRFRenderer::Render forwards a pose to), main.cpp:203-213 dimensions
//   adapter_check bench train_hash    [steps] [rays] [optimizer]   the reference's loop body (NeRFExecutor.h:862-995) on the drop-in: Optimizer->zero_grad(); Render on the
//   adapter_check bench train_classic [steps] [rays] [optimizer]   ray batch; mse_loss + huber_loss; loss.backward(); Optimizer->step() -- torch::optim::Adam over the modules'
//   adapter_check bench train_lerf    [steps] [rays] [optimizer]   own parameters (:508-539); optimizer = adam (torch::optim::Adam, the reference's) | hipadam (nrfpp::HipAdam)
//
// What a C++ host of the reference gets from the drop-in, on the host's clock: the frame calls go through the reference's own virtual Render() signature, the training
// steps through the statements of NeRFExecutor::Train.  Scenes are the ones bench.py times through the Python mirror (nerfpp_amd/scene.py: same closed-form weights,
// include/nrf_synth.h), so that `dump_dir` can hold the first frame for a bit-for-bit comparison with the mirror's.  One JSON line per run.
//
// Timing: a settling phase (whole frames / steps until three in a row are within 10 % of the fastest seen), then the timed calls between two device synchronisations,
// nothing synchronised in between except what the reference's own statements synchronise (Near / Far read-back in Render, the loss print of the LeRF branch).  A second,
// shorter pass runs with nrfpp::phase_clock() on: every phase is then drained on entry and exit, so its time is its own host + device work.
#include "adapter_util.h"
#include "LeRF.h"

#include <c10/hip/HIPGuard.h>

#include <chrono>
#include <iostream>
#include <sstream>
#include <unistd.h>

static const char *train_gemm_name()
{
	switch (nrf_get_train_gemm()) { case 0: return "f32"; case 1: return "bf16x3"; case 2: return "f16x3"; default: return "by family (classic, LeRF: f16x3; NeRFSmall fp32 backward: f32)"; }
}


using Clock = std::chrono::steady_clock;
static double ms_since(Clock::time_point t0) { return std::chrono::duration<double, std::milli>(Clock::now() - t0).count(); }
static void dev_sync() { c10::hip::getCurrentHIPStream().synchronize(); torch::cuda::synchronize(); }

// the 3 * L hash primes of nerfpp_amd/scene.py::CU_PRIMES: nrf_synth_u32(424242, i) % (2^30 - 2^28) + 2^28, kept when prime
static std::vector<int32_t> cu_primes(int count)
{
	std::vector<int32_t> out;
	for (uint32_t i = 0; (int)out.size() < count; i++) {
		const uint32_t v = nrf_synth_u32(424242u, i) % ((1u << 30) - (1u << 28)) + (1u << 28);
		bool prime = true;
		for (uint32_t q = 2; (uint64_t)q * q <= v; q++) if (v % q == 0) { prime = false; break; }
		if (prime) out.push_back((int32_t)v);
	}
	return out;
}

static std::string json_map(const std::map<std::string, double> &m, double scale = 1.0)
{
	std::ostringstream o; o << "{"; bool first = true;
	for (auto &kv : m) { if (!first) o << ", "; first = false; char b[64]; snprintf(b, sizeof b, "%.4f", kv.second * scale); o << "\"" << kv.first << "\": " << b; }
	o << "}"; return o.str();
}

/// runs `fn` (one synchronised frame / step) until three in a row are within 10 % of the fastest one and `floor_s` has passed (at most `budget_s`)
template <class F> static void settle(F fn, double floor_s = 1.5, double budget_s = 10.0)
{
	double best = 1e30; int good = 0; const auto t_begin = Clock::now();
	for (;;) {
		dev_sync(); const auto t0 = Clock::now();
		fn(); dev_sync();
		const double dt = ms_since(t0);
		best = std::min(best, dt);
		good = dt <= 1.10 * best ? good + 1 : 0;
		const double el = ms_since(t_begin) * 1e-3;
		if ((good >= 3 && el >= floor_s) || el >= budget_s) break;
	}
}

// ---- scenes (nerfpp_amd/scene.py) ------------------------------------------------------------------------------------------------------------------------------
using HashRenderer = nrfpp::HipNeRFRenderer<nrfpp::HipHashEmbedder, nrfpp::HipSHEncoder, NeRFSmall>;
using ClassicRenderer = nrfpp::HipNeRFRenderer<nrfpp::HipEmbedder, nrfpp::HipEmbedder, ClassicModel>;

struct HashScene {
	torch::Tensor bbox = torch::tensor({-1.5f, -1.5f, -1.5f, 1.5f, 1.5f, 1.5f});
	nrfpp::HipHashEmbedder e{nullptr};
	nrfpp::HipSHEncoder ed{nullptr};
	NeRFSmall m{nullptr};
	std::unique_ptr<HashRenderer> r;
	HashScene(float table_amp, float sigma_scale, int precision, int log2_t = 19)      // make_hash_scene(mode="cu", table_amp, sigma_scale, log2_t)
	{
		const int L = 16, F = 2, T = log2_t;
		e = nrfpp::HipHashEmbedder("embedder", bbox, L, F, T, 16, 512, NRF_HASH_CU);
		{
			torch::NoGradGuard ng;
			const int64_t per_level = ((int64_t)1 << T) * F;
			auto flat = e->Embeddings.view({-1});
			for (int l = 0; l < L; l++) fill_synth(flat.narrow(0, l * per_level, per_level), 5000u + 1000u * l, table_amp);
			auto pr = cu_primes(3 * L);
			e->SetPrimes(torch::from_blob(pr.data(), {L, 1, 3}, torch::kInt32).clone());
		}
		ed = nrfpp::HipSHEncoder("embeddirs", 3, 4, NRF_SH_CUDA);
		m = NeRFSmall(3, 64, 15, 4, 64, false, 3, 64, L * F, 16, "model");
		int k = 0;
		for (auto &p : m->named_parameters()) {
			auto t = p.value();
			float amp = 1.6f * std::sqrt(6.0f / float(t.size(0) + t.size(1)));
			if (p.key().find("sigma_net_2") != std::string::npos) amp = amp * sigma_scale;
			fill_synth(t, 6000u + 1000u * (k++), amp);
		}
		m->to(torch::kCUDA);
		e->Initialize();
		r = std::make_unique<HashRenderer>(e, ed, m, precision);
		nrf_mlp_small_desc sd{L * F, 16, 3, 64, 15, 4, 64};
		r->SyncWeights(&sd, nullptr);
	}
};

struct ClassicScene {
	torch::Tensor bbox = torch::tensor({-1.5f, -1.5f, -1.5f, 1.5f, 1.5f, 1.5f});
	nrfpp::HipEmbedder e{nullptr}, ed{nullptr};
	ClassicModel m{nullptr};
	std::unique_ptr<ClassicRenderer> r;
	explicit ClassicScene(int precision)                                 // make_classic_scene()
	{
		e = nrfpp::HipEmbedder("embedder", 10); ed = nrfpp::HipEmbedder("embeddirs", 4);
		m = ClassicModel(8, 256, 63, 27, 5, std::set<int>{4}, true, "model");
		int k = 0;
		for (auto &p : m->named_parameters()) {
			auto t = p.value();
			float amp = t.dim() == 2 ? 1.4f * std::sqrt(6.0f / float(t.size(0) + t.size(1))) : 0.1f;
			if (p.key().find("alpha_linear.weight") != std::string::npos) amp = amp * 40.0f;
			fill_synth(t, 7000u + 1000u * (k++), amp);
		}
		m->to(torch::kCUDA);
		r = std::make_unique<ClassicRenderer>(e, ed, m, precision);
		nrf_mlp_nerf_desc cd{8, 256, 63, 27, 5, 4, 1};
		r->SyncWeights(nullptr, &cd);
	}
};

struct LeRFScene {
	torch::Tensor bbox = torch::tensor({-1.5f, -1.5f, -1.5f, 1.5f, 1.5f, 1.5f});
	nrfpp::HipHashEmbedder le{nullptr};
	LeRF lerf{nullptr};
	std::unique_ptr<nrfpp::HipLeRFPass> pass;
	LeRFScene()                                                          // make_lerf_scene(): main.cpp:203-213
	{
		const int L = 16, F = 8, T = 19;
		le = nrfpp::HipHashEmbedder("lang_embedder", bbox, L, F, T, 16, 1024, NRF_HASH_CU);
		{
			torch::NoGradGuard ng;
			fill_synth(le->Embeddings.view({-1}), 311u, 0.5f);
			auto pr = cu_primes(3 * L);
			le->SetPrimes(torch::from_blob(pr.data(), {L, 1, 3}, torch::kInt32).clone());
		}
		le->Initialize();
		lerf = LeRF(32, 2, 256, 768, L * F, "lang_model");
		int k = 0;
		for (auto &p : lerf->named_parameters()) {
			auto t = p.value();
			float amp = 1.6f * std::sqrt(6.0f / float(t.size(0) + t.size(1)));
			if (p.key().find("sigma_le_net_1") != std::string::npos) amp = amp * 20.0f;
			fill_synth(t, 1311u + 1000u * (k++), amp);
		}
		lerf->to(torch::kCUDA);
		pass = std::make_unique<nrfpp::HipLeRFPass>(le, NRF_PREC_F16_SPLIT);
		pass->SyncWeights(lerf);
	}
};

static NeRFRenderParams lego_params(torch::Tensor bbox, int chunk, bool white, bool weights, bool raw)
{
	NeRFRenderParams rp;
	rp.NSamples = 64; rp.NImportance = 128; rp.Chunk = chunk; rp.ReturnRaw = raw; rp.LinDisp = false; rp.Perturb = 0.f; rp.WhiteBkgr = white; rp.RawNoiseStd = 0.f;
	rp.Ndc = false; rp.UseViewdirs = true; rp.ReturnWeights = weights; rp.ThinRay = true; rp.RenderFactor = 0; rp.BoundingBox = bbox.cuda(); rp.StochasticPreconditioningAlpha = 0.f;
	return rp;
}

// ---- frames ----------------------------------------------------------------------------------------------------------------------------------------------------
template <class RenderFrame>
static void time_frames(const char *family, const char *surface, RenderFrame render, int poses, const std::vector<int> &lane_counts, const std::string &extra)
{
	std::map<std::string, double> by_lanes, host_by_lanes;
	int best_lanes = 0; double best = 1e30;
	for (int lanes : lane_counts) {
		nrfpp::check(nrf_set_render_lanes(lanes), "nrf_set_render_lanes");
		render(0);                                                        // a new lane count sizes the workspace
		settle([&] { render(0); });
		double host = 0.0;
		dev_sync(); const auto t0 = Clock::now();
		for (int k = 0; k < poses; k++) { const auto th = Clock::now(); render(k); host += ms_since(th); }
		dev_sync();
		const double ms = ms_since(t0) / poses;
		by_lanes[std::to_string(lanes)] = ms; host_by_lanes[std::to_string(lanes)] = host / poses;
		if (ms < best) { best = ms; best_lanes = lanes; }
	}
	printf("{\"what\": \"dropin_frame\", \"family\": \"%s\", \"surface\": \"%s\", \"image\": [800, 800], \"samples\": \"64+128\", \"poses\": %d, \"ms_per_frame\": %.4f, \"lanes\": %d, "
		"\"ms_per_frame_by_lanes\": %s, \"host_ms_inside_render_by_lanes\": %s, \"value\": %.6g, \"unit\": \"ray-samples/s\"%s}\n", family, surface, poses, best, best_lanes,
		json_map(by_lanes).c_str(), json_map(host_by_lanes).c_str(), 640000.0 * 256.0 / (best * 1e-3), extra.c_str());
	fflush(stdout);
}

static int bench_frame_hash(int poses, const char *dump_dir)
{
	torch::NoGradGuard ng;
	HashScene sc(0.5f, 30.0f, NRF_PREC_F16_SPLIT);
	auto rp = lego_params(sc.bbox, 65536, true, false, false);
	auto K = lego_K(800, 800);
	std::vector<torch::Tensor> c2w;
	for (int k = 0; k < poses; k++) c2w.push_back(orbit_pose(-180.f + 9.f * k, -30.f, 4.f));
	NeRFRenderer<nrfpp::HipHashEmbedder, nrfpp::HipSHEncoder, NeRFSmall> *base = sc.r.get();     // the call goes through the reference's virtual, as NeRFExecutor::RenderView makes it
	torch::Tensor first;
	auto render = [&](int k) { auto res = base->Render(800, 800, K, rp, {torch::Tensor(), torch::Tensor(), torch::Tensor()}, c2w[k]); if (k == 0) first = res.Outputs.RGBMap; };
	std::string extra;
	if (dump_dir) {
		render(0); dev_sync();
		write_f32(std::string(dump_dir) + "/dropin_frame_hash_rgb.f32", first);
		extra = ", \"dumped\": \"dropin_frame_hash_rgb.f32\"";
	}
	time_frames("hash", "HipNeRFRenderer<HipHashEmbedder, HipSHEncoder, NeRFSmall>::Render via NeRFRenderer<>* (f16x3)", render, poses, {1, 2}, extra);
	return 0;
}

static int bench_frame_classic(int poses)
{
	torch::NoGradGuard ng;
	ClassicScene sc(NRF_PREC_F16_SPLIT);
	auto rp = lego_params(sc.bbox, 8192, true, false, false);
	auto K = lego_K(800, 800);
	std::vector<torch::Tensor> c2w;
	for (int k = 0; k < poses; k++) c2w.push_back(orbit_pose(-180.f + 9.f * k, -30.f, 4.f));
	ClassicRendererBase *base = nullptr;
	NeRFRenderer<nrfpp::HipEmbedder, nrfpp::HipEmbedder, ClassicModel> *b2 = sc.r.get(); (void)base;
	auto render = [&](int k) { b2->Render(800, 800, K, rp, {torch::Tensor(), torch::Tensor(), torch::Tensor()}, c2w[k]); };
	time_frames("classic", "HipNeRFRenderer<HipEmbedder, HipEmbedder, NeRF 8x256>::Render via NeRFRenderer<>* (f16x3, exact-fp32 coarse density)", render, poses, {2}, "");
	return 0;
}

static int bench_frame_lerf(int poses)
{
	torch::NoGradGuard ng;
	LeRFScene sc;
	auto posp = torch::nn::functional::normalize(torch::randn({1, 768}), torch::nn::functional::NormalizeFuncOptions().dim(-1));
	auto negp = torch::nn::functional::normalize(torch::randn({3, 768}), torch::nn::functional::NormalizeFuncOptions().dim(-1));
	sc.pass->SetLeRFPrompts(posp, negp);
	auto K = lego_K(800, 800);
	std::vector<torch::Tensor> c2w;
	for (int k = 0; k < poses; k++) c2w.push_back(orbit_pose(-180.f + 9.f * k, -30.f, 4.f));
	auto render = [&](int k) { float nr, fr; sc.pass->Render(800, 800, K, sc.bbox, 64, 128, 32768, c2w[k], true, false, true, &nr, &fr); };
	time_frames("lerf", "HipLeRFPass::Render (HipLeRFRenderer::Render's pose branch; weights + rendered embedding + Relevancy returned)", render, poses, {1}, "");
	return 0;
}

// ---- training steps --------------------------------------------------------------------------------------------------------------------------------------------
struct StepClock {
	std::map<std::string, double> host;       // unsynchronised: the host's own time inside each statement
	void add(const char *k, Clock::time_point t0) { host[k] += ms_since(t0); }
};

static std::unique_ptr<torch::optim::Optimizer> make_optimizer(const std::string &kind, std::vector<torch::Tensor> vars, double lr)
{
	if (kind == "hipadam") return std::make_unique<nrfpp::HipAdam>(vars, torch::optim::AdamOptions(lr).eps(1e-15).betas(std::make_tuple(0.9, 0.99)));
	return std::make_unique<torch::optim::Adam>(vars, torch::optim::AdamOptions(lr).eps(1e-15).betas(std::make_tuple(0.9, 0.99)));      // NeRFExecutor.h:539
}

/// `body(clock)` is one iteration of NeRFExecutor::Train's loop (returns the loss tensor); timed unsynchronised over `steps`, then `phase_steps` times with the phase clock on
template <class Body>
static void time_steps(const char *family, const char *surface, Body body, int steps, int64_t rays, const std::string &optimizer, const std::string &extra)
{
	StepClock warm;
	float loss_first = body(warm).template item<float>();
	// a FIXED number of warm-up steps (not the frame benches' time-based settle): every body() is an optimizer step, so the loss the timed steps end at -- and, with the
	// random targets of the LeRF bench, whether the run has already reached the trivial optimum (all weights zero: rendered embedding 0, loss 0.5, degenerate
	// gradients, 25 % slower steps) -- must not depend on how many steps fit into a second on this box
	const bool show = getenv("NRF_BENCH_LOSSES") != nullptr;          // (debugging: the loss of every warm-up step on stderr)
	if (show) fprintf(stderr, "loss %.6g", loss_first);
	for (int i = 0; i < 8; i++) { auto l = body(warm); if (show) fprintf(stderr, " %.6g", l.template item<float>()); }
	if (show) fprintf(stderr, "\n");
	dev_sync();
	StepClock sc;
	torch::Tensor loss;
	dev_sync(); const auto t0 = Clock::now();
	for (int i = 0; i < steps; i++) loss = body(sc);
	dev_sync();
	const double ms = ms_since(t0) / steps;
	const float loss_last = loss.template item<float>();
	// per-phase, each drained on entry and exit
	const int phase_steps = std::max(2, std::min(steps, 5));
	nrfpp::phase_clock().reset(); nrfpp::phase_clock().On = true;
	StepClock scp;
	for (int i = 0; i < phase_steps; i++) body(scp);
	nrfpp::phase_clock().On = false;
	printf("{\"what\": \"dropin_train_step\", \"family\": \"%s\", \"surface\": \"%s\", \"optimizer\": \"%s\", \"rays_per_step\": %lld, \"samples\": \"64+128\", \"steps\": %d, \"ms_per_step\": %.4f, "
		"\"value\": %.6g, \"unit\": \"ray-samples/s\", \"train_gemm\": \"%s\", \"host_ms_per_statement\": %s, \"synchronised_phase_ms\": %s, \"loss_first_last\": [%.6g, %.6g]%s}\n", family, surface, optimizer.c_str(),
		(long long)rays, steps, ms, (double)rays * 256.0 / (ms * 1e-3), train_gemm_name(), json_map(sc.host, 1.0 / steps).c_str(), json_map(nrfpp::phase_clock().Ms, 1.0 / phase_steps).c_str(), loss_first,
		loss_last, extra.c_str());
	fflush(stdout);
}

/// a ray batch of the Lego camera: `n` rays spread over the 800 x 800 frame of pose (30, -30, 4)
static std::pair<torch::Tensor, torch::Tensor> ray_batch(int64_t n)
{
	auto [ro, rd, cone] = GetRays(800, 800, lego_K(800, 800), orbit_pose(30.f, -30.f, 4.f));
	auto idx = torch::arange(0, n, torch::kLong) * (640000 / n);
	return {ro.reshape({-1, 3}).index_select(0, idx).contiguous().cuda(), rd.reshape({-1, 3}).index_select(0, idx).contiguous().cuda()};
}

template <class Renderer, class Embedder_, class Model>
static void nerf_train_steps(const char *family, const char *surface, Renderer *NeRFRenderer, Embedder_ &e, Model &m, torch::Tensor bbox, int steps, int64_t n_rand,
	const std::string &optimizer)
{
	struct { struct { torch::Tensor rays_o, rays_d, cone_angle; } data; struct { torch::Tensor target_s; } target; } batch;
	std::tie(batch.data.rays_o, batch.data.rays_d) = ray_batch(n_rand);
	torch::manual_seed(7);
	batch.target.target_s = torch::rand({n_rand, 3}).cuda();
	std::vector<torch::Tensor> grad_vars;                                 // NeRFExecutor.h:508-535: embedder first, then the model
	for (auto &p : e->parameters()) grad_vars.push_back(p);
	for (auto &p : m->parameters()) grad_vars.push_back(p);
	auto Optimizer = make_optimizer(optimizer, grad_vars, 5e-4);
	auto rp = lego_params(bbox, (int)n_rand, false, true, false);          // FillRenderParams at train time: ReturnWeights, ReturnRaw = false (main.cpp), one Chunk per batch
	auto *render_params = &rp;
	const torch::Device Device(torch::kCUDA);
	auto body = [&](StepClock &c) {
		nrfpp::PhaseScope whole("step");
		auto t = Clock::now();
		{ nrfpp::PhaseScope ps("zero_grad"); Optimizer->zero_grad(); }                                                            // :866
		c.add("zero_grad", t); t = Clock::now();
		torch::Tensor loss = torch::full({1}, 0.f).to(Device), psnr = torch::full({1}, 10.f).to(Device);                           // :868-869
		auto rgb_disp_acc_extras = [&] { nrfpp::PhaseScope ps("render"); return NeRFRenderer->Render(0, 0, torch::Tensor(), *render_params,
			{batch.data.rays_o, batch.data.rays_d, batch.data.cone_angle}, torch::Tensor(), torch::Tensor()); }();                     // :876
		c.add("render", t); t = Clock::now();
		torch::Tensor mse_loss, img_loss;
		{
			nrfpp::PhaseScope ps("loss");
			mse_loss = torch::mse_loss(rgb_disp_acc_extras.Outputs.RGBMap, batch.target.target_s.detach());                         // :882
			img_loss = torch::nn::functional::huber_loss(rgb_disp_acc_extras.Outputs.RGBMap, batch.target.target_s.detach());       // :883
			loss = img_loss;
			torch::NoGradGuard no_grad;
			psnr = -10. * torch::log(mse_loss) / torch::log(torch::full({1}, 10.f)).to(Device);                                     // :893
		}
		c.add("loss", t); t = Clock::now();
		{ nrfpp::PhaseScope ps("backward"); loss.backward(); }                                                                      // :923
		c.add("backward", t); t = Clock::now();
		{ nrfpp::PhaseScope ps("optimizer_step"); Optimizer->step(); }                                                              // :985
		c.add("optimizer_step", t);
		return loss;
	};
	time_steps(family, surface, body, steps, n_rand, optimizer, "");
}

static int bench_train_hash(int steps, int64_t n_rand, const std::string &optimizer)
{
	HashScene sc(1e-2f, 4.0f, NRF_PREC_F16_SPLIT);                       // bench.py's hashnerf_train_step scene
	NeRFRenderer<nrfpp::HipHashEmbedder, nrfpp::HipSHEncoder, NeRFSmall> *base = sc.r.get();
	nerf_train_steps("hash", "NeRFExecutor::Train's loop body on HipNeRFRenderer<HipHashEmbedder, HipSHEncoder, NeRFSmall> (f16x3 render)", base, sc.e, sc.m, sc.bbox, steps, n_rand, optimizer);
	return 0;
}

static int bench_train_classic(int steps, int64_t n_rand, const std::string &optimizer)
{
	ClassicScene sc(NRF_PREC_F16_SPLIT);
	NeRFRenderer<nrfpp::HipEmbedder, nrfpp::HipEmbedder, ClassicModel> *base = sc.r.get();
	nerf_train_steps("classic", "NeRFExecutor::Train's loop body on HipNeRFRenderer<HipEmbedder, HipEmbedder, NeRF 8x256> (f16x3 render)", base, sc.e, sc.m, sc.bbox, steps, n_rand, optimizer);
	return 0;
}

struct LeRFBenchRenderer {                 // LeRFRenderer::Render's signature (LeRFRenderer.h:125-132) in front of the pass HipLeRFRenderer forwards a ray batch to
	nrfpp::HipLeRFPass &Pass;
	struct Result { nrfpp::LeRFPassOutputs Outputs; torch::Tensor Raw; float Near = 0.f, Far = 0.f; };
	Result Render(const int h, const int w, torch::Tensor k, const NeRFRenderParams &p, std::tuple<torch::Tensor, torch::Tensor, torch::Tensor> rays, torch::Tensor c2w, torch::Tensor c2w_staticcam)
	{
		Result r;
		r.Outputs = Pass.RenderBatch(std::get<0>(rays), std::get<1>(rays), p.BoundingBox, p.NSamples, p.NImportance, p.Chunk, p.LinDisp, p.ReturnWeights, &r.Near, &r.Far);
		return r;
	}
	LeRFBenchRenderer *operator->() { return this; }
};

static int bench_train_lerf(int steps, int64_t n_rand, const std::string &optimizer)
{
	LeRFScene sc;
	LeRFBenchRenderer LeRFRenderer{*sc.pass};
	struct { struct { torch::Tensor rays_o, rays_d, cone_angle; } data; struct { torch::Tensor target_lang_embedding; } target; } batch;
	std::tie(batch.data.rays_o, batch.data.rays_d) = ray_batch(n_rand);
	torch::manual_seed(77);
	batch.target.target_lang_embedding = torch::nn::functional::normalize(torch::randn({n_rand, 768}), torch::nn::functional::NormalizeFuncOptions().dim(-1)).cuda();
	std::vector<torch::Tensor> grad_vars;
	for (auto &p : sc.le->parameters()) grad_vars.push_back(p);
	for (auto &p : sc.lerf->parameters()) grad_vars.push_back(p);
	auto Optimizer = make_optimizer(optimizer, grad_vars, 5e-4);
	auto rp = lego_params(sc.bbox, 32768, false, true, false);
	rp.UseViewdirs = false;
	auto *render_params = &rp;
	torch::Tensor loss = torch::full({1}, 0.f).cuda();
	std::ostringstream sink;
	auto body = [&](StepClock &c) {
		nrfpp::PhaseScope whole("step");
		auto t = Clock::now();
		{ nrfpp::PhaseScope ps("zero_grad"); Optimizer->zero_grad(); }
		c.add("zero_grad", t); t = Clock::now();
		auto lerf_render_result = [&] { nrfpp::PhaseScope ps("render"); return LeRFRenderer->Render(0, 0, torch::Tensor(), *render_params,
			{batch.data.rays_o, batch.data.rays_d, batch.data.cone_angle}, torch::Tensor(), torch::Tensor()); }();                     // :958-961
		c.add("render", t); t = Clock::now();
		torch::Tensor lang_loss;
		{
			nrfpp::PhaseScope ps("loss");
			lang_loss = torch::nn::functional::huber_loss(lerf_render_result.Outputs.RenderedLangEmbedding.to(loss.device()), batch.target.target_lang_embedding.detach().to(loss.device()),
				torch::nn::functional::HuberLossFuncOptions().reduction(torch::kNone).delta(1.25)).sum(-1).nanmean();                    // :970-974
			sink.str(""); sink << "lang_loss: " << lang_loss << std::endl;                                                            // :980 (the print reads the loss back: a host synchronisation per step)
		}
		c.add("loss", t); t = Clock::now();
		{ nrfpp::PhaseScope ps("backward"); lang_loss.backward(); }                                                                  // :981
		c.add("backward", t); t = Clock::now();
		{ nrfpp::PhaseScope ps("optimizer_step"); Optimizer->step(); }                                                               // :985
		c.add("optimizer_step", t);
		return lang_loss;
	};
	time_steps("lerf", "NeRFExecutor::Train's LeRF branch on HipLeRFPass::RenderBatch (HipLeRFRenderer::Render's ray-batch branch; f16x3 render)", body, steps, n_rand, optimizer, "");
	return 0;
}

// ---- adapter_check train_dp: data-parallel training of the drop-in at world 2, the ranks being two THREADS of this process over the RCCL named by NRF_RCCL_LIBRARY
// (tests: tests/helpers/mock_rccl.cpp -- RCCL itself refuses two ranks on one device).  Each rank: its own replica (modules, renderer, optimizer, stream) and its own
// half of every ray batch; the loop body of NeRFExecutor::Train with ONE added statement between loss.backward() and Optimizer->step():
// comm.AllReduceGrads(grad_vars) (nrfpp::TileComm -> nrf_allreduce_grads).  After the steps the replicas must hold the same parameters bit for bit; a third replica
// trained WITHOUT the exchange on rank 0's half must not (the exchange happened).
#include <thread>

static int run_train_dp_impl(const std::string &optimizer)
{
	const int world = 2, steps = 3; const int64_t n = 1024;
	auto [ro, rd] = ray_batch(world * n);
	torch::manual_seed(11);
	auto target = torch::rand({world * n, 3}).cuda();
	char idp[128]; snprintf(idp, sizeof idp, "/tmp/nrf_train_dp_comm_%ld", (long)getpid());
	const std::string id_path = idp;
	std::vector<std::vector<torch::Tensor>> finals(world + 1);
	std::vector<std::string> errors(world + 1);
	auto rank_main = [&](int rank, bool exchange) {
		try {
			c10::hip::HIPStreamGuard sg(c10::hip::getStreamFromPool());
			HashScene sc(1e-2f, 4.0f, NRF_PREC_F16_SPLIT, 14);
			std::unique_ptr<nrfpp::TileComm> comm;
			if (exchange) comm = std::make_unique<nrfpp::TileComm>(world, rank, id_path, 60.0, "train_dp");
			std::vector<torch::Tensor> grad_vars;
			for (auto &p : sc.e->parameters()) grad_vars.push_back(p);
			for (auto &p : sc.m->parameters()) grad_vars.push_back(p);
			auto Optimizer = make_optimizer(optimizer, grad_vars, 1e-2);
			auto rp = lego_params(sc.bbox, (int)n, false, true, false);
			auto o = ro.narrow(0, rank * n, n).contiguous(), d = rd.narrow(0, rank * n, n).contiguous(), tg = target.narrow(0, rank * n, n).contiguous();
			for (int i = 0; i < steps; i++) {
				Optimizer->zero_grad();
				auto res = sc.r->Render(0, 0, torch::Tensor(), rp, {o, d, torch::Tensor()}, torch::Tensor(), torch::Tensor());
				auto loss = torch::nn::functional::huber_loss(res.Outputs.RGBMap, tg.detach());
				loss.backward();
				if (comm) comm->AllReduceGrads(grad_vars);                   // the one added statement
				Optimizer->step();
			}
			c10::hip::getCurrentHIPStream().synchronize();
			for (auto &p : grad_vars) finals[exchange ? rank : world].push_back(p.detach().clone().cpu());
		} catch (const std::exception &ex) { errors[exchange ? rank : world] = ex.what(); }
	};
	std::vector<std::thread> th;
	for (int r = 0; r < world; r++) th.emplace_back(rank_main, r, true);
	for (auto &t : th) t.join();
	rank_main(0, false);
	std::string note;
	for (auto &e : errors) if (!e.empty()) { note += e.substr(0, 160) + "; "; }
	for (auto &ch : note) if (ch == '"' || ch == '\n') ch = ' ';
	bool same = note.empty() && finals[0].size() == finals[1].size() && !finals[0].empty(), differs_without = false;
	double moved = 0.0;
	if (note.empty()) {
		for (size_t i = 0; same && i < finals[0].size(); i++) same = torch::equal(finals[0][i], finals[1][i]);
		for (size_t i = 0; i < finals[0].size() && i < finals[world].size(); i++) if (!torch::equal(finals[0][i], finals[world][i])) differs_without = true;
		moved = (finals[0].back() - finals[world].back()).abs().max().item<double>();
	}
	const bool ok = same && differs_without;
	printf("{\"train_dp_ok\": %s, \"world\": %d, \"steps\": %d, \"rays_per_rank\": %lld, \"optimizer\": \"%s\", \"replicas_bit_identical\": %s, \"differs_from_a_replica_without_the_exchange\": %s, "
		"\"last_layer_max_abs_diff_vs_unsynchronised\": %.3e, \"note\": \"%s\"}\n", ok ? "true" : "false", world, steps, (long long)n, optimizer.c_str(), same ? "true" : "false",
		differs_without ? "true" : "false", moved, note.c_str());
	fflush(stdout);
	return ok ? 0 : 1;
}

int run_train_dp(int argc, const char **argv)
{
	if (!torch::cuda::is_available()) { printf("{\"train_dp_ok\": false, \"error\": \"no GPU\"}\n"); return 2; }
	std::streambuf *cout_buf = std::cout.rdbuf();
	std::ostringstream quiet;
	std::cout.rdbuf(quiet.rdbuf());
	const int rc = run_train_dp_impl(argc > 0 ? argv[0] : "adam");
	std::cout.rdbuf(cout_buf);
	fflush(stdout);
	_exit(rc);          // (a LibTorch-HIP process that has created communicators aborts in the runtimes' exit handlers: adapter_check.cpp, end of main)
}

int run_bench(int argc, const char **argv)
{
	if (!torch::cuda::is_available()) { printf("{\"what\": \"dropin\", \"error\": \"no GPU\"}\n"); return 2; }
	const std::string what = argc > 0 ? argv[0] : "";
	std::streambuf *cout_buf = std::cout.rdbuf();
	std::ostringstream quiet;
	std::cout.rdbuf(quiet.rdbuf());                                       // Trainable::Initialize and friends print parameter names
	int rc = 2;
	try {
		const int a1 = argc > 1 ? atoi(argv[1]) : 0;
		if (what == "frame_hash") rc = bench_frame_hash(a1 > 0 ? a1 : 10, argc > 2 ? argv[2] : nullptr);
		else if (what == "frame_classic") rc = bench_frame_classic(a1 > 0 ? a1 : 4);
		else if (what == "frame_lerf") rc = bench_frame_lerf(a1 > 0 ? a1 : 5);
		else {
			const int64_t rays = argc > 2 ? atoll(argv[2]) : 0;
			const std::string opt = argc > 3 ? argv[3] : "adam";
			if (what == "train_hash") rc = bench_train_hash(a1 > 0 ? a1 : 20, rays > 0 ? rays : 16384, opt);
			else if (what == "train_classic") rc = bench_train_classic(a1 > 0 ? a1 : 5, rays > 0 ? rays : 4096, opt);
			else if (what == "train_lerf") rc = bench_train_lerf(a1 > 0 ? a1 : 5, rays > 0 ? rays : 16384, opt);
			else fprintf(stderr, "adapter_check bench frame_hash|frame_classic|frame_lerf|train_hash|train_classic|train_lerf ...\n");
		}
	} catch (const std::exception &ex) {
		std::string note = ex.what(); for (auto &ch : note) if (ch == '"' || ch == '\n') ch = ' ';
		printf("{\"what\": \"dropin\", \"bench\": \"%s\", \"error\": \"%s\"}\n", what.c_str(), note.substr(0, 400).c_str());
		rc = 1;
	}
	std::cout.rdbuf(cout_buf);
	fflush(stdout);
	return rc;
}

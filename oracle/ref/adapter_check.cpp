// adapter_check.cpp -- TEST INFRASTRUCTURE (built only where /root/reference exists; the binary travels in oracle/_ref/).
//
// The drop-in demonstration, inside the reference's own template machinery:
//   reference   NeRFRenderer<HashEmbedder, SHEncoder, NeRFSmall>             on LibTorch CPU          (the oracle)
//   this repo   nrfpp::HipNeRFRenderer<HipHashEmbedder, HipSHEncoder, NeRFSmall> : NeRFRenderer<...>  on the MI355X
// Both are driven through the reference's unmodified Render() -> BatchifyRays() -> (virtual) RenderRays() and given the same
// synthetic weights.  Against the LibTorch CPU render >= 90 % of the pixels must agree within 1e-4 and the render-vs-render PSNR must
// exceed 55 dB: the reference's fine-pass sample set is a discontinuous function of the coarse weights, and the reference differs
// from ITSELF by this much between CPU dispatch settings ("How exact can parity be").  The adapter's matrix-core precision
// (NRF_PREC_F16_SPLIT) against its own NRF_PREC_F32 render -- which equals the CPU oracle bit for bit -- is held to the strict bar:
// every pixel value within 1e-4.  The multi-GPU surface (RenderTile, RenderSharded over a TileComm) is exercised on a world of one.
// Also checks the BaseEmbedder surface
// (GetOutputDims / forward) of the hash and SH encoders against the reference modules.
//
// usage: adapter_check [h w]      prints one JSON line, exit code 0 iff every check passed
#define NRFPP_WITH_REFERENCE
#include "nerfpp_torch.h"
#include "nrf_synth.h"
#include "LeRF.h"
#include "LeRFRenderer.h"       // RenderCLIPEmbedding (inline, LeRFRenderer.h:45-54); the LeRFRenderer class itself is never instantiated here (its unit needs RuCLIP)

#include <cstdio>
#include <iostream>
#include <unistd.h>

using torch::indexing::Slice;
using torch::indexing::None;

static void fill_synth(torch::Tensor p, uint32_t seed, float amp)
{
	torch::NoGradGuard ng;
	auto flat = torch::empty({p.numel()}, torch::kFloat32);
	float *d = flat.data_ptr<float>();
	for (int64_t i = 0; i < p.numel(); i++) d[i] = nrf_synth_sym(seed, (uint32_t)i, amp);
	p.copy_(flat.view(p.sizes()));
}

static torch::Tensor lego_K(int h, int w)
{
	float focal = 0.5f * w / std::tan(0.5f * 0.6911112f);
	float kdata[] = {focal, 0, 0.5f * w, 0, focal, 0.5f * h, 0, 0, 1};
	return torch::from_blob(kdata, {3, 3}).clone();
}

static torch::Tensor orbit_pose(float theta_deg, float phi_deg, float radius)
{
	const float PI_ = std::acos(-1.0f);
	float th = theta_deg / 180.f * PI_, ph = phi_deg / 180.f * PI_;
	float t_[] = {1,0,0,0, 0,1,0,0, 0,0,1,radius, 0,0,0,1};
	float rp[] = {1,0,0,0, 0,std::cos(ph),-std::sin(ph),0, 0,std::sin(ph),std::cos(ph),0, 0,0,0,1};
	float rt[] = {std::cos(th),0,-std::sin(th),0, 0,1,0,0, std::sin(th),0,std::cos(th),0, 0,0,0,1};
	float fl[] = {-1,0,0,0, 0,0,1,0, 0,1,0,0, 0,0,0,1};
	auto c2w = torch::from_blob(t_, {4,4}).clone();
	c2w = torch::matmul(torch::from_blob(rp, {4,4}).clone(), c2w);
	c2w = torch::matmul(torch::from_blob(rt, {4,4}).clone(), c2w);
	c2w = torch::matmul(torch::from_blob(fl, {4,4}).clone(), c2w);
	return c2w.index({Slice(None, 3), Slice(None, 4)}).contiguous();
}

static void stage(const char *what) { if (getenv("NRF_ADAPTER_TRACE")) { fprintf(stderr, "[adapter_check] %s\n", what); fflush(stderr); } }

int main(int argc, const char **argv)
{
	const int h = argc > 1 ? atoi(argv[1]) : 16, w = argc > 2 ? atoi(argv[2]) : 16;
	if (!torch::cuda::is_available()) { printf("{\"ok\": false, \"error\": \"no GPU\"}\n"); return 2; }
	std::streambuf *cout_buf = std::cout.rdbuf();
	torch::NoGradGuard ng;
	const int L = 16, F = 2, T = 17;      // 2^17 rows per level keeps the CPU table small; semantics identical
	auto bbox = torch::tensor({-1.5f, -1.5f, -1.5f, 1.5f, 1.5f, 1.5f});
	// ---- the reference's modules (CPU) with synthetic weights ----
	HashEmbedder e("embedder", bbox, L, F, T, 16, 512);
	SHEncoder ed("embeddirs", 3, 4);
	NeRFSmall m(3, 64, 15, 4, 64, false, 3, 64, L * F, 16, "model");
	int k = 0;
	for (auto &p : e->named_parameters()) fill_synth(p.value(), 5000u + 1000u * (k++), 0.5f);
	k = 0;
	for (auto &p : m->named_parameters()) {
		auto t = p.value();
		float amp = 1.6f * std::sqrt(6.0f / float(t.size(0) + t.size(1)));
		if (p.key().find("sigma_net_2") != std::string::npos) amp *= 8.0f;
		fill_synth(t, 6000u + 1000u * (k++), amp);
	}
	// ---- the same model behind the HIP adapters ----
	nrfpp::HipHashEmbedder he("embedder", bbox, L, F, T, 16, 512, NRF_HASH_NGP);
	bool names_equal = true;
	{
		// same parameter names as the reference module (NeRF.cpp:255-259: `embedder_embeddings_<i>.weight`), copied level by level
		auto pr = e->named_parameters(); auto ph = he->named_parameters();
		names_equal = pr.size() == ph.size();
		for (size_t i = 0; names_equal && i < pr.size(); i++) {
			names_equal = pr[i].key() == ph[i].key() && pr[i].value().sizes() == ph[i].value().sizes();
			if (names_equal) ph[i].value().copy_(pr[i].value());
		}
		he->Sync();
	}
	nrfpp::HipSHEncoder hd("embeddirs", 3, 4, NRF_SH_LIBTORCH);
	nrfpp::HipNeRFRenderer<nrfpp::HipHashEmbedder, nrfpp::HipSHEncoder, NeRFSmall> hip(he, hd, m, NRF_PREC_F32);
	nrf_mlp_small_desc sd{L * F, 16, 3, 64, 15, 4, 64};
	hip.SyncWeights(&sd, nullptr);

	bool ok = true;
	// ---- BaseEmbedder surface ----
	auto x = (torch::rand({4096, 3}) * 3.2f - 1.6f);
	auto [emb_ref, mask_ref] = e->forward(x);
	auto [emb_hip, mask_hip] = he->forward(x.cuda());
	const bool emb_exact = torch::equal(emb_hip.cpu(), emb_ref) && torch::equal(mask_hip.cpu(), mask_ref);
	ok = ok && emb_exact && he->GetOutputDims() == e->GetOutputDims();
	auto dirs = torch::nn::functional::normalize(torch::randn({1024, 3}), torch::nn::functional::NormalizeFuncOptions().dim(-1));
	const bool sh_exact = torch::equal(hd->forward(dirs.cuda()).first.cpu(), ed->forward(dirs).first);
	ok = ok && sh_exact;

	// ---- Render(): reference CPU vs HIP, through the reference's own Render/BatchifyRays ----
	NeRFRenderParams rp;
	rp.NSamples = 64; rp.NImportance = 128; rp.Chunk = 100; rp.ReturnRaw = false; rp.LinDisp = false; rp.Perturb = 0.f; rp.WhiteBkgr = true;
	rp.RawNoiseStd = 0.f; rp.Ndc = false; rp.UseViewdirs = true; rp.ReturnWeights = true; rp.ThinRay = true; rp.BoundingBox = bbox;
	auto K = lego_K(h, w);
	auto c2w = orbit_pose(30.f, -30.f, 4.f);
	NeRFRenderer<HashEmbedder, SHEncoder, NeRFSmall> ref(e, ed, m);
	auto r_ref = ref.Render(h, w, K, rp, {torch::Tensor(), torch::Tensor(), torch::Tensor()}, c2w);
	auto rp_gpu = rp; rp_gpu.BoundingBox = bbox.cuda();
	auto r_hip = hip.Render(h, w, K.cuda(), rp_gpu, {torch::Tensor(), torch::Tensor(), torch::Tensor()}, c2w.cuda());
	const float rgb_err = (r_hip.Outputs.RGBMap.cpu() - r_ref.Outputs.RGBMap).abs().max().item<float>();
	const float acc_err = (r_hip.Outputs.AccMap.cpu() - r_ref.Outputs.AccMap).abs().max().item<float>();
	const float dep_err = (r_hip.Outputs.DepthMap.cpu() - r_ref.Outputs.DepthMap).abs().max().item<float>();
	// The fine-pass sample set is a discontinuous function of the coarse weights (searchsorted on CDF plateaus): ulp-level
	// differences between MKL/SLEEF on the CPU and the HIP path move a few samples, so a few pixels differ by more than the
	// typical 1e-5.  Reported: the share of pixels within 1e-4 and the PSNR, next to the max.
	const float frac_1e4 = ((r_hip.Outputs.RGBMap.cpu() - r_ref.Outputs.RGBMap).abs().amax(-1) < 1e-4f).to(torch::kFloat32).mean().item<float>();
	const double mse = (r_hip.Outputs.RGBMap.cpu() - r_ref.Outputs.RGBMap).pow(2).mean().item<double>();
	const double psnr = mse > 0 ? -10.0 * std::log10(mse) : 999.0;
	const bool shape_ok = r_hip.Outputs.RGBMap.sizes() == r_ref.Outputs.RGBMap.sizes() && r_hip.Outputs.DepthMap.sizes() == r_ref.Outputs.DepthMap.sizes() &&
		r_hip.Near == r_ref.Near && r_hip.Far == r_ref.Far;
	ok = ok && frac_1e4 >= 0.90f && psnr > 55.0 && shape_ok;
	// fast precision through the same surface
	hip.SetPrecision(NRF_PREC_F16_MFMA);
	auto r_f16 = hip.Render(h, w, K.cuda(), rp_gpu, {torch::Tensor(), torch::Tensor(), torch::Tensor()}, c2w.cuda());
	const bool f16_finite = torch::isfinite(r_f16.Outputs.RGBMap).all().item<bool>();
	ok = ok && f16_finite;
	// split precision: the FAST path (dense pyramid, level-major hi/lo features, matrix-core MLP) behind the same reference surface
	hip.SetPrecision(NRF_PREC_F16_SPLIT);
	auto r_sp = hip.Render(h, w, K.cuda(), rp_gpu, {torch::Tensor(), torch::Tensor(), torch::Tensor()}, c2w.cuda());
	const float split_frac_1e4 = ((r_sp.Outputs.RGBMap.cpu() - r_ref.Outputs.RGBMap).abs().amax(-1) < 1e-4f).to(torch::kFloat32).mean().item<float>();
	const double split_mse = (r_sp.Outputs.RGBMap.cpu() - r_ref.Outputs.RGBMap).pow(2).mean().item<double>();
	const double split_psnr = split_mse > 0 ? -10.0 * std::log10(split_mse) : 999.0;
	ok = ok && split_frac_1e4 >= 0.90f && split_psnr > 55.0;
	// ... and against the adapter's OWN NRF_PREC_F32 render (== the CPU oracle bit for bit): the split mode's coarse pass evaluates the sigma net in exact fp32,
	// so the sample sets coincide and EVERY pixel value is within 1e-4 (strict; the comparison with the LibTorch CPU render above carries MKL's summation order)
	const float split_vs_f32 = (r_sp.Outputs.RGBMap - r_hip.Outputs.RGBMap).abs().max().item<float>();
	ok = ok && split_vs_f32 < 1e-4f;
	stage("single-GPU renders done");
	// the library's chunk loop (nrf_render_rows, the default) against the reference's own BatchifyRays driving the virtual RenderRays: same pixels, bit for bit
	hip.LibraryChunkLoop = false;
	auto r_loop = hip.Render(h, w, K.cuda(), rp_gpu, {torch::Tensor(), torch::Tensor(), torch::Tensor()}, c2w.cuda());
	hip.LibraryChunkLoop = true;
	const bool chunk_loop_same = torch::equal(r_loop.Outputs.RGBMap, r_sp.Outputs.RGBMap) && torch::equal(r_loop.Outputs.DepthMap, r_sp.Outputs.DepthMap) &&
		torch::equal(r_loop.Outputs.Weights, r_sp.Outputs.Weights) && r_loop.Near == r_sp.Near && r_loop.Far == r_sp.Far;
	// Ndc + UseViewdirs and c2w_staticcam through the fused Render (the reference's own Ndc render reads a dangling `sh`, NeRFRenderer.h:562/567, so the comparison
	// is with the reference's pieces: GetRays -> viewdirs -> NDCRays -> IntersectWithAABB on the CPU, and the explicit-ray-batch branch of the adapter)
	bool ndc_ok = false, static_ok = false;
	{
		auto c2n = torch::tensor({{0.98f, -0.05f, 0.19f, 0.10f}, {0.06f, 0.995f, -0.07f, -0.05f}, {-0.185f, 0.08f, 0.98f, 0.20f}});
		auto rpn = rp_gpu; rpn.Ndc = true; rpn.Chunk = 77;
		auto r_n = hip.Render(h, w, K.cuda(), rpn, {torch::Tensor(), torch::Tensor(), torch::Tensor()}, c2n.cuda());
		auto [ro, rd, cone] = GetRays(h, w, K, c2n);
		auto r_b = hip.Render(h, w, K.cuda(), rpn, {ro.reshape({-1, 3}).cuda(), rd.reshape({-1, 3}).cuda(), cone});
		ndc_ok = r_n.Outputs.RGBMap.sizes() == std::vector<int64_t>({h, w, 3}) && torch::isfinite(r_n.Outputs.RGBMap).all().item<bool>() &&
			torch::equal(r_n.Outputs.RGBMap.reshape({-1, 3}), r_b.Outputs.RGBMap) && r_n.Near == r_b.Near && r_n.Far == r_b.Far;
		auto [no, nd, nc] = NDCRays(h, w, K[0][0].item<float>(), 1.f, ro, rd, torch::Tensor());
		auto [nr, fr] = IntersectWithAABB(no.reshape({-1, 3}), nd.reshape({-1, 3}), bbox, 0.f);
		ndc_ok = ndc_ok && r_n.Near == nr.min().item<float>() && r_n.Far == fr.max().item<float>();
		auto c2s = orbit_pose(-20.f, -35.f, 3.6f);
		auto r_s = hip.Render(h, w, K.cuda(), rp_gpu, {torch::Tensor(), torch::Tensor(), torch::Tensor()}, c2w.cuda(), c2s.cuda());
		auto r_ref_s = ref.Render(h, w, K, rp, {torch::Tensor(), torch::Tensor(), torch::Tensor()}, c2w, c2s);
		const float frac_s = ((r_s.Outputs.RGBMap.cpu() - r_ref_s.Outputs.RGBMap).abs().amax(-1) < 1e-4f).to(torch::kFloat32).mean().item<float>();
		static_ok = frac_s >= 0.90f && r_s.Near == r_ref_s.Near && r_s.Far == r_ref_s.Far;
	}
	ok = ok && names_equal && chunk_loop_same && ndc_ok && static_ok;
	stage("chunk loop / NDC / staticcam done");
	// ---- module state of the drop-in embedders (CuHashEmbedder.cpp:24-76, NeRF.cpp:255-271) ----
	bool cu_scratch_ok = false, load_cu_ok = false, load_ngp_ok = false, saved = false, zero_primes_rejected = false;
	std::string state_note = "ok";
	try {
		const int SL = 4, SF = 2, ST = 12;
		torch::manual_seed(1234);
		nrfpp::HipHashEmbedder cu("embedder", bbox, SL, SF, ST, 16, 128, NRF_HASH_CU);       // constructed from scratch, exactly as INTEGRATION.md section 1 says
		cu->Initialize();
		auto pr = cu->Primes.cpu().reshape({-1});
		bool primes_ok = pr.numel() == 3 * SL;
		for (int64_t i = 0; primes_ok && i < pr.numel(); i++) {
			const int v = pr[i].item<int>();
			primes_ok = v >= (1 << 28) && v < (1 << 30);
			for (int q = 2; primes_ok && (int64_t)q * q <= v; q++) if (v % q == 0) primes_ok = false;
		}
		// the same generator state gives the reference's draw sequence: rand for the table first, then one randint per candidate prime
		torch::manual_seed(1234);
		auto tab_ref = torch::rand({((int64_t)1 << ST) * SL, SF}, torch::TensorOptions().dtype(torch::kFloat32).device(torch::kCUDA)) * 1e-4f;
		const bool table_ok = torch::equal(tab_ref, cu->Embeddings.detach());
		auto xs = (torch::rand({2048, 3}) * 2.8f - 1.4f).cuda();
		auto emb_s = cu->forward(xs).first;                                                   // zero primes would hash every corner to row 0: identical rows
		const bool spread = (emb_s.std(0) > 0).all().item<bool>() && (emb_s.amax(0) - emb_s.amin(0)).min().item<float>() > 1e-6f;
		auto names = cu->named_buffers();
		const bool buffers_ok = names.size() == 4 && names.contains("embedder_feat_local_size") && names.contains("embedder_feat_local_idx") &&
			names["embedder_feat_local_size"].dtype() == torch::kInt32 && names["embedder_feat_local_idx"][SL - 1].item<int>() == (SL - 1) * (1 << ST) &&
			names["embedder_primes"].sizes() == std::vector<int64_t>({SL, 1, 3}) && names["embedder_biases"].sizes() == std::vector<int64_t>({SL, 3});
		cu_scratch_ok = primes_ok && table_ok && spread && buffers_ok;
		{
			nrf_hash *hh = nullptr; nrf_hash_desc hd{NRF_HASH_CU, SL, SF, ST, 16, 128, {-1.5f, -1.5f, -1.5f, 1.5f, 1.5f, 1.5f}};
			nrfpp::check(nrf_hash_create(&hd, &hh), "nrf_hash_create");
			std::vector<int32_t> z(3 * SL, 0);
			zero_primes_rejected = nrf_hash_set_primes(hh, z.data(), nullptr) == NRF_ERR_INVALID_ARG;
			nrf_hash_destroy(hh);
		}
		const char *ck = getenv("NRF_ADAPTER_CKPT_DIR"), *outd = getenv("NRF_ADAPTER_OUT_DIR");
		if (ck) {
			// the reference-written fixtures (tests/golden/ckpt, ref_driver ckpt_save) straight into the adapter modules with torch::load
			torch::load(cu, std::string(ck) + "/cu_embedder_checkpoint.pt");
			cu->to(torch::kCUDA);                                                               // as the executor does after its loads (NeRFExecutor.h:552-556)
			cu->Sync();
			auto p2 = cu->Primes.cpu().reshape({-1});
			load_cu_ok = p2[0].item<int>() == 268435459 && p2[5].item<int>() == 268435469 && (cu->Biases.cpu() == 0.25f).all().item<bool>() &&
				torch::isfinite(cu->forward(xs).first).all().item<bool>();
			nrfpp::HipHashEmbedder ng("embedder", bbox, SL, SF, ST, 16, 128, NRF_HASH_NGP);
			torch::load(ng, std::string(ck) + "/embedder_checkpoint.pt");
			ng->Sync();                                                                         // (left on the CPU on purpose: Sync uploads from wherever the parameters are)
			HashEmbedder e_ref("embedder", bbox, SL, SF, ST, 16, 128);
			torch::load(e_ref, std::string(ck) + "/embedder_checkpoint.pt");
			auto xc = xs.cpu();
			load_ngp_ok = torch::equal(ng->forward(xs).first.cpu(), e_ref->forward(xc).first);
			if (outd) {
				// files the adapters torch::save: ref_driver ckpt_load (the reference's own modules + torch::load) must accept them (checked by the caller)
				torch::save(ng, std::string(outd) + "/embedder_checkpoint.pt");
				torch::save(cu, std::string(outd) + "/cu_embedder_checkpoint.pt");
				saved = true;
			}
		}
	} catch (const std::exception &ex) { state_note = ex.what(); for (auto &ch : state_note) if (ch == '"' || ch == '\n') ch = ' '; state_note = state_note.substr(0, 300); }
	ok = ok && cu_scratch_ok && zero_primes_rejected && (!getenv("NRF_ADAPTER_CKPT_DIR") || (load_cu_ok && load_ngp_ok));
	stage("module state done");
	// ---- multi-GPU surface: RenderTile == the slice of Render, RenderSharded over a world of one == Render (the box has one GPU) ----
	const int row0 = h / 3, rows = h / 2;
	auto r_tile = hip.RenderTile(h, w, K.cuda(), rp_gpu, c2w.cuda(), row0, rows);
	const bool tile_exact = torch::equal(r_tile.Outputs.RGBMap, r_sp.Outputs.RGBMap.index({Slice(row0, row0 + rows)})) &&
		torch::equal(r_tile.Outputs.DepthMap, r_sp.Outputs.DepthMap.index({Slice(row0, row0 + rows)}));
	stage("RenderTile done");
	bool sharded_exact = false;
	std::string comm_note = "ok";
	if (getenv("NRF_ADAPTER_SKIP_COMM")) { sharded_exact = true; comm_note = "skipped"; }
	else try {
		const std::string id_path = std::string("/tmp/nrf_adapter_check_comm_") + std::to_string((long)getpid());
		nrfpp::TileComm comm(1, 0, id_path);
		auto r_sh = hip.RenderSharded(h, w, K.cuda(), rp_gpu, c2w.cuda(), comm);
		sharded_exact = torch::equal(r_sh.Outputs.RGBMap, r_sp.Outputs.RGBMap) && torch::equal(r_sh.Outputs.DepthMap, r_sp.Outputs.DepthMap) &&
			torch::equal(r_sh.Outputs.AccMap.reshape({-1}), r_sp.Outputs.AccMap.reshape({-1})) && r_sh.Near == r_sp.Near && r_sh.Far == r_sp.Far;
	} catch (const std::exception &ex) { comm_note = ex.what(); }
	stage("RenderSharded done");
	ok = ok && tile_exact && sharded_exact;
	// ---- LeRF render pass (BASELINE config 4): nrfpp::HipLeRFPass -- what HipLeRFRenderer : LeRFRenderer forwards to -- against the reference's own LeRF module
	// (LeRF.cpp, LibTorch CPU) and RenderCLIPEmbedding (LeRFRenderer.h:45-54) on the same sample points.  The language hash grid is CUDA-only in the
	// reference (CuHashEmbedder), so its features come from the HIP encoder on both sides.
	bool lerf_ok = false, lerf_fused = false, lerf_reuse_same = false;
	double lerf_cos_min = 0.0, lerf_w_err = 1.0, lerf_f16_cos_min = 0.0;
	std::string lerf_note = "ok";
	try {
		const int LL = 16, LF = 8, LT = 14;
		nrfpp::HipHashEmbedder le("lang_embedder", bbox, LL, LF, LT, 16, 256, NRF_HASH_CU);
		fill_synth(le->Embeddings.data(), 311u, 0.5f);
		{
			std::vector<int32_t> pr;
			for (int32_t c = 268435459; (int)pr.size() < 3 * LL; c += 2) { bool is_p = true; for (int32_t q = 3; (int64_t)q * q <= c; q += 2) if (c % q == 0) { is_p = false; break; } if (is_p) pr.push_back(c); }
			le->SetPrimes(torch::from_blob(pr.data(), {LL, 1, 3}, torch::kInt32).clone());
		}
		le->Initialize();
		LeRF lerf(32, 2, 256, 768, LL * LF, "lang_model");
		k = 0;
		for (auto &p : lerf->named_parameters()) {
			auto t = p.value();
			float amp = 1.6f * std::sqrt(6.0f / float(t.size(0) + t.size(1)));
			if (p.key().find("sigma_le_net_1") != std::string::npos) amp *= 20.0f;
			fill_synth(t, 8000u + 1000u * (k++), amp);
		}
		stage("LeRF modules built");
		nrfpp::HipLeRFPass pass(le, NRF_PREC_F16_SPLIT);
		pass.SyncWeights(lerf);
		stage("LeRF SyncWeights done");
		lerf_fused = pass.IsFused();
		// a ray batch of the same camera
		auto ro = torch::empty({(int64_t)h * w, 3}, torch::TensorOptions().dtype(torch::kFloat32).device(torch::kCUDA)), rd = torch::empty_like(ro);
		auto Kh = nrfpp::host_floats(K), Mh = nrfpp::host_floats(c2w);
		nrfpp::check(nrf_get_rays(h, w, Kh.data(), Mh.data(), 0, h, ro.data_ptr<float>(), rd.data_ptr<float>(), nullptr, nrfpp::current_stream()), "nrf_get_rays");
		auto bbh = nrfpp::host_floats(bbox);
		auto rays_ = torch::empty({(int64_t)h * w, 11}, ro.options());
		nrfpp::check(nrf_pack_rays(ro.data_ptr<float>(), rd.data_ptr<float>(), bbh.data(), (int64_t)h * w, 1, rays_.data_ptr<float>(), nrfpp::current_stream()), "nrf_pack_rays");
		torch::Tensor zf;
		auto got = pass.RenderRays(rays_, 64, false, 128, true, &zf);
		stage("LeRF fused RenderRays done");
		// the feature-reusing passes (default) against the two plain passes: same kernels on the same inputs
		pass.ReuseFeatures = false;
		torch::Tensor zf2;
		auto plain = pass.RenderRays(rays_, 64, false, 128, true, &zf2);
		pass.ReuseFeatures = true;
		// both see the exact-fp32 coarse pass, hence the same sample set; the reusing passes take the coarse depths' sigma_le from that exact pass instead of
		// re-evaluating it in split arithmetic: weights agree to the split precision's own level
		lerf_reuse_same = torch::equal(zf, zf2) && (got.WeightsLE - plain.WeightsLE).abs().max().item<double>() < 1e-5 * plain.WeightsLE.abs().max().item<double>() &&
			(got.RenderedLangEmbedding - plain.RenderedLangEmbedding).abs().max().item<double>() < 2e-5;
		// the reference side, on the fused pass's own fine depths
		auto rc = rays_.cpu(); auto zc = zf.cpu();
		auto pts = rc.index({Slice(), None, Slice(0, 3)}) + rc.index({Slice(), None, Slice(3, 6)}) * zc.index({Slice(), Slice(), None});
		auto [emb_l, keep_l] = le->forward(pts.reshape({-1, 3}).cuda());
		auto raw = lerf->forward(emb_l.cpu());                                                  // LeRFImpl::forward, LibTorch CPU
		raw.index_put_({~keep_l.cpu(), -1}, 0);                                                 // LeRFRenderer.cpp:22-23
		raw = raw.view({(int64_t)h * w, 192, 769});
		auto st_out = pass.RawToLEOutputs(raw.cuda(), zf, rays_.index({Slice(), Slice(3, 6)}).contiguous(), 768);      // weights: the library's fp32 stage (oracle-pinned)
		auto emb_ref = RenderCLIPEmbedding(raw.index({"...", Slice(0, 768)}), st_out.WeightsLE.cpu().unsqueeze(-1));       // the reference's own function
		stage("LeRF reference side done");
		auto hit = st_out.AccMapLE.cpu() > 1e-2f;
		auto cosv = (got.RenderedLangEmbedding.cpu() * emb_ref).sum(-1).index({hit});
		lerf_cos_min = cosv.numel() ? cosv.min().item<double>() : 0.0;
		lerf_w_err = (got.WeightsLE.cpu() - st_out.WeightsLE.cpu()).abs().max().item<double>();
		nrfpp::HipLeRFPass pass16(le, NRF_PREC_F16_MFMA);
		pass16.SyncWeights(lerf);
		auto got16 = pass16.RenderRays(rays_, 64, false, 128, true);
		auto cos16 = (got16.RenderedLangEmbedding.cpu() * emb_ref).sum(-1).index({hit});
		lerf_f16_cos_min = cos16.numel() ? cos16.min().item<double>() : 0.0;
		lerf_ok = lerf_fused && lerf_reuse_same && hit.sum().item<int64_t>() > (int64_t)h * w / 8 && lerf_cos_min > 1.0 - 2e-6 && lerf_w_err < 1e-5 && torch::isfinite(got.RenderedLangEmbedding).all().item<bool>();
	} catch (const std::exception &ex) { lerf_note = ex.what(); for (auto &ch : lerf_note) if (ch == '"' || ch == '\n') ch = ' '; }
	stage("LeRF section done");
	ok = ok && lerf_ok;
	std::cout.rdbuf(cout_buf);
	printf("{\"lerf_pass_ok\": %s, \"lerf_fused\": %s, \"lerf_feature_reuse_equals_two_passes\": %s, \"lerf_split_cos_min_vs_reference_head\": %.9f, \"lerf_split_weights_max_abs_err\": %.3e, \"lerf_f16_cos_min\": %.6f, \"lerf_note\": \"%s\"}\n",
		lerf_ok ? "true" : "false", lerf_fused ? "true" : "false", lerf_reuse_same ? "true" : "false", lerf_cos_min, lerf_w_err, lerf_f16_cos_min, lerf_note.c_str());
	printf("{\"module_state_ok\": %s, \"parameter_names_equal_reference\": %s, \"cu_from_scratch_primes_table_buffers_ok\": %s, \"zero_primes_rejected\": %s, \"torch_load_cu_fixture\": %s, "
		"\"torch_load_ngp_fixture_forward_bit_exact\": %s, \"adapter_checkpoints_saved\": %s, \"chunk_loop_library_equals_reference_batchify\": %s, \"ndc_viewdirs_ok\": %s, "
		"\"staticcam_ok\": %s, \"state_note\": \"%s\"}\n", (cu_scratch_ok && zero_primes_rejected) ? "true" : "false", names_equal ? "true" : "false", cu_scratch_ok ? "true" : "false",
		zero_primes_rejected ? "true" : "false", load_cu_ok ? "true" : "false", load_ngp_ok ? "true" : "false", saved ? "true" : "false", chunk_loop_same ? "true" : "false",
		ndc_ok ? "true" : "false", static_ok ? "true" : "false", state_note.c_str());
	printf("{\"ok\": %s, \"image\": [%d, %d], \"hash_embedding_bit_exact\": %s, \"sh_bit_exact\": %s, \"rgb_max_abs_err\": %.3e, \"pixels_within_1e-4\": %.4f, \"acc_max_abs_err\": %.3e, "
		"\"depth_max_abs_err\": %.3e, \"psnr_db\": %.2f, \"shapes_near_far_equal\": %s, \"f16_render_finite\": %s, \"split_pixels_within_1e-4\": %.4f, \"split_psnr_db\": %.2f, "
		"\"split_vs_own_f32_max_abs_err\": %.3e, \"render_tile_equals_slice\": %s, \"render_sharded_world1_equals_render\": %s, \"comm\": \"%s\"}\n",
		ok ? "true" : "false", h, w, emb_exact ? "true" : "false", sh_exact ? "true" : "false", rgb_err, frac_1e4, acc_err, dep_err, psnr, shape_ok ? "true" : "false",
		f16_finite ? "true" : "false", split_frac_1e4, split_psnr, split_vs_f32, tile_exact ? "true" : "false", sharded_exact ? "true" : "false", comm_note.c_str());
	fflush(stdout);
	stage("results printed");
	// A LibTorch-HIP process that has initialised RCCL (torch's bundled 2.26.6 here) aborts inside the runtimes' exit handlers ("double free or corruption"),
	// with or without ncclCommDestroy -- measured; the Python hosts are not affected.  Everything owned here is already released: leave without running them.
	if (comm_note != "skipped") { fflush(stderr); _exit(ok ? 0 : 1); }
	return ok ? 0 : 1;
}

// ref_driver.cpp -- TEST INFRASTRUCTURE (not product code).
//
// Drives the *reference's own* LibTorch CPU path (compiled from the sources where
// they lie under /root/reference/src by oracle/build_ref.sh) to
//   (1) emit golden input/output vectors for every hot-path stage  ("golden <dir>")
//   (2) time the reference CPU renderer for bench.py's cpu_baseline  ("bench ...").
//
// Nothing in here restates reference arithmetic: every number written to a golden
// file is produced by a function or class of the reference (RayUtils.h, Sampler.h,
// NeRF.{h,cpp}, CustomOps.{h,cpp}, LeRF.{h,cpp}, NeRFRenderer.h), except the few
// arrays explicitly prefixed "aux_" which are derived with the same ATen ops in the
// same order as the cited reference lines, to expose values the reference keeps local.
//
// Model parameters are NOT random: they are filled from include/nrf_synth.h so the
// tests can regenerate them on the GPU box without shipping multi-megabyte blobs.

#include "NeRF.h"
#include "LeRF.h"
#include "Sampler.h"
#include "CustomOps.h"
#include "NeRFRenderer.h"   // filtered copy (OpenCV image helpers removed), see build_ref.sh
#include "LeRFRenderer.h"   // for the inline RenderCLIPEmbedding (:45-54); LeRFRenderer.cpp itself needs RuCLIP and is not built

#include "nrf_synth.h"

#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <iostream>
#include <map>

using torch::indexing::Slice;
using torch::indexing::None;

// ----------------------------------------------------------------------------------------------
// .npy writer (v1.0, C order)
// ----------------------------------------------------------------------------------------------
static std::string g_outdir;
static std::ofstream g_manifest;

static void save_npy(const std::string &name, torch::Tensor t)
{
	t = t.detach().cpu().contiguous();
	std::string descr;
	if (t.scalar_type() == torch::kFloat32) descr = "<f4";
	else if (t.scalar_type() == torch::kFloat64) descr = "<f8";
	else if (t.scalar_type() == torch::kInt64) descr = "<i8";
	else if (t.scalar_type() == torch::kInt32) descr = "<i4";
	else if (t.scalar_type() == torch::kBool) descr = "|b1";
	else if (t.scalar_type() == torch::kUInt8) descr = "|u1";
	else { std::cerr << "save_npy: unsupported dtype for " << name << std::endl; std::exit(2); }
	std::ostringstream shape;
	shape << "(";
	for (int64_t i = 0; i < t.dim(); i++) { shape << t.size(i) << ","; }
	shape << ")";
	std::string hdr = "{'descr': '" + descr + "', 'fortran_order': False, 'shape': " + shape.str() + ", }";
	size_t total = 10 + hdr.size() + 1;
	size_t pad = (64 - total % 64) % 64;
	hdr += std::string(pad, ' ');
	hdr += "\n";
	std::ofstream f(g_outdir + "/" + name + ".npy", std::ios::binary);
	const char magic[] = "\x93NUMPY";
	f.write(magic, 6);
	char ver[2] = {1, 0};
	f.write(ver, 2);
	uint16_t hl = (uint16_t)hdr.size();
	f.write(reinterpret_cast<char*>(&hl), 2);
	f.write(hdr.data(), hdr.size());
	f.write(reinterpret_cast<const char*>(t.data_ptr()), t.numel() * t.element_size());
}

static void save_scalar(const std::string &name, double v) { save_npy(name, torch::tensor({(float)v})); }

// ----------------------------------------------------------------------------------------------
// synthetic parameters (include/nrf_synth.h); a manifest line per tensor lets the tests
// rebuild exactly the same tensors in numpy.
// ----------------------------------------------------------------------------------------------
static void fill_synth(torch::Tensor p, uint32_t seed, float amp)
{
	torch::NoGradGuard ng;
	auto flat = torch::empty({p.numel()}, torch::kFloat32);
	float *d = flat.data_ptr<float>();
	for (int64_t i = 0; i < p.numel(); i++)
		d[i] = nrf_synth_sym(seed, (uint32_t)i, amp);
	p.copy_(flat.view(p.sizes()));
}

// gain<0 : absolute amplitude |gain|;  else xavier-uniform-like amplitude gain*sqrt(6/(fan_in+fan_out))
template <class M>
static void fill_module(const std::string &model_tag, M &module, uint32_t base_seed, float w_gain, float bias_amp,
	const std::map<std::string, float> &per_tensor_scale = {})
{
	int k = 0;
	for (auto &p : module->named_parameters())
	{
		auto t = p.value();
		float amp;
		if (t.dim() == 2 && p.key().find("embeddings") == std::string::npos)
			amp = w_gain * std::sqrt(6.0f / float(t.size(0) + t.size(1)));
		else if (t.dim() == 2)
			amp = w_gain;		//embedding tables: w_gain is the absolute amplitude
		else
			amp = bias_amp;
		for (auto &kv : per_tensor_scale)
			if (p.key().find(kv.first) != std::string::npos)
				amp *= kv.second;
		uint32_t seed = base_seed + 1000u * (uint32_t)k;
		fill_synth(t, seed, amp);
		g_manifest << model_tag << " " << p.key() << " " << seed << " " << std::setprecision(9) << amp;
		for (int64_t i = 0; i < t.dim(); i++) g_manifest << " " << t.size(i);
		g_manifest << "\n";
		k++;
	}
}

static torch::Tensor synth_tensor(std::vector<int64_t> shape, uint32_t seed, float amp, float offset = 0.f)
{
	int64_t n = 1; for (auto s : shape) n *= s;
	auto t = torch::empty({n}, torch::kFloat32);
	float *d = t.data_ptr<float>();
	for (int64_t i = 0; i < n; i++) d[i] = nrf_synth_sym(seed, (uint32_t)i, amp) + offset;
	return t.view(shape);
}

// ----------------------------------------------------------------------------------------------
// inputs: Lego-shaped camera. Input generators only -- nothing here is graded.
// ----------------------------------------------------------------------------------------------
static torch::Tensor lego_K(int h, int w)
{
	const float cax = 0.6911112f;
	float focal = 0.5f * w / std::tan(0.5f * cax);
	float kdata[] = { focal, 0, 0.5f * w, 0, focal, 0.5f * h, 0, 0, 1 };
	return torch::from_blob(kdata, {3, 3}).clone();
}

static torch::Tensor orbit_pose(float theta_deg, float phi_deg, float radius)
{
	// camera on a sphere looking at the origin, blender convention; an input generator
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

// ----------------------------------------------------------------------------------------------
// Spy: the reference renderer with its protected virtuals opened up and every
// intermediate recorded. All arithmetic is Base::.
// ----------------------------------------------------------------------------------------------
template <class E, class D, class M>
struct Spy : public NeRFRenderer<E, D, M>
{
	using Base = NeRFRenderer<E, D, M>;
	Spy(E e, D d, M m) : Base(e, d, m) {}
	std::vector<torch::Tensor> net_pts, net_raw, r2o_raw, r2o_z, r2o_d, batch_rays;
	std::vector<NeRFRendererOutputs> r2o_out;
	std::vector<torch::Tensor> raw_live;		//the graph tensors themselves (retain_grad), for the training-step goldens

	torch::Tensor RunNetwork(torch::Tensor inputs, torch::Tensor view_dirs, M fn, E embed_fn, D embeddirs_fn) override
	{
		auto out = Base::RunNetwork(inputs, view_dirs, fn, embed_fn, embeddirs_fn);
		net_pts.push_back(inputs.detach().clone());
		net_raw.push_back(out.detach().clone());
		return out;
	}
	NeRFRendererOutputs RawToOutputs(torch::Tensor raw, torch::Tensor cone_angle, torch::Tensor z_vals, torch::Tensor rays_d,
		const float raw_noise_std = 0.f, const bool white_bkgr = false) override
	{
		if (raw.requires_grad()) { raw.retain_grad(); raw_live.push_back(raw); }
		auto out = Base::RawToOutputs(raw, cone_angle, z_vals, rays_d, raw_noise_std, white_bkgr);
		r2o_raw.push_back(raw.detach().clone());
		r2o_z.push_back(z_vals.detach().clone());
		r2o_d.push_back(rays_d.detach().clone());
		r2o_out.push_back(out);
		return out;
	}
	NeRFRenderResult BatchifyRays(torch::Tensor rays_flat, torch::Tensor cone_angle, const int n_samples, const int chunk = 1024 * 32,
		const bool return_raw = false, const bool lin_disp = false, const float perturb = 0.f, const int n_importance = 0,
		const bool white_bkgr = false, const float raw_noise_std = 0., const float spa = 0.f,
		torch::Tensor bounding_box = torch::Tensor(), const bool return_weights = true) override
	{
		batch_rays.push_back(rays_flat.detach().clone());
		return Base::BatchifyRays(rays_flat, cone_angle, n_samples, chunk, return_raw, lin_disp, perturb, n_importance, white_bkgr,
			raw_noise_std, spa, bounding_box, return_weights);
	}
	NeRFRendererOutputs OpenRawToOutputs(torch::Tensor raw, torch::Tensor z_vals, torch::Tensor rays_d, bool white)
	{
		return Base::RawToOutputs(raw, torch::Tensor(), z_vals, rays_d, 0.f, white);
	}
};

static void save_outputs(const std::string &prefix, const NeRFRendererOutputs &o)
{
	if (o.RGBMap.defined()) save_npy(prefix + "rgb", o.RGBMap);
	if (o.DispMap.defined()) save_npy(prefix + "disp", o.DispMap);
	if (o.AccMap.defined()) save_npy(prefix + "acc", o.AccMap);
	if (o.Weights.defined()) save_npy(prefix + "weights", o.Weights);
	if (o.DepthMap.defined()) save_npy(prefix + "depth", o.DepthMap);
}

static torch::Tensor lego_bbox() { return torch::tensor({-1.5f, -1.5f, -1.5f, 1.5f, 1.5f, 1.5f}); }

static NeRFRenderParams lego_params(int ns, int ni, int chunk)
{
	NeRFRenderParams rp;
	rp.NSamples = ns; rp.NImportance = ni; rp.Chunk = chunk;
	rp.ReturnRaw = true; rp.LinDisp = false; rp.Perturb = 0.f; rp.WhiteBkgr = true; rp.RawNoiseStd = 0.f;
	rp.Ndc = false; rp.UseViewdirs = true; rp.ReturnWeights = true; rp.ThinRay = true; rp.RenderFactor = 0;
	rp.BoundingBox = lego_bbox();
	rp.StochasticPreconditioningAlpha = 0.f;
	return rp;
}

// ----------------------------------------------------------------------------------------------
// golden groups
// ----------------------------------------------------------------------------------------------
static void g_rays()
{
	// R1: GetDirections + GetRays (RayUtils.h:5-46), non-square to pin row-major (y outer, x inner)
	const int h = 6, w = 10;
	auto k = lego_K(h, w);
	auto c2w = orbit_pose(30.f, -30.f, 4.f);
	auto [o, d, cone] = GetRays(h, w, k, c2w);
	save_npy("rays.k", k); save_npy("rays.c2w", c2w);
	save_npy("rays.hw", torch::tensor({h, w}, torch::kInt32));
	save_npy("rays.o", o.contiguous()); save_npy("rays.d", d); save_npy("rays.cone", cone.reshape({1}));
	// R2: NDCRays (RayUtils.h:49-83) on the same rays
	auto [no, nd, nc] = NDCRays(h, w, k[0][0].item<float>(), 1.f, o, d, cone);
	save_npy("rays.ndc_o", no); save_npy("rays.ndc_d", nd); save_npy("rays.ndc_cone", nc);
	// second camera: 8x8 Lego view used by the render goldens
	const int h2 = 8, w2 = 8;
	auto k2 = lego_K(h2, w2);
	auto c2 = orbit_pose(-63.f, -30.f, 4.f);
	auto [o2, d2, cone2] = GetRays(h2, w2, k2, c2);
	save_npy("rays.k2", k2); save_npy("rays.c2w2", c2);
	save_npy("rays.o2", o2.contiguous()); save_npy("rays.d2", d2); save_npy("rays.cone2", cone2.reshape({1}));
}

static void g_aabb()
{
	// R3: IntersectWithAABB (RayUtils.h:87-126)
	auto bbox = lego_bbox();
	auto o = synth_tensor({48, 3}, 11u, 4.0f);
	auto d = synth_tensor({48, 3}, 12u, 1.0f);
	{
		// hand-placed edge cases: axis-parallel, zero and -1e-6 components, origin inside the box, rays pointing away
		auto oa = o.accessor<float, 2>(); auto da = d.accessor<float, 2>();
		oa[0][0] = 0; oa[0][1] = 0; oa[0][2] = 4;  da[0][0] = 0; da[0][1] = 0; da[0][2] = -1;
		oa[1][0] = 0; oa[1][1] = 0; oa[1][2] = 4;  da[1][0] = 0; da[1][1] = 0; da[1][2] = 1;		//points away
		oa[2][0] = 0.2f; oa[2][1] = -0.3f; oa[2][2] = 0.1f;  da[2][0] = 0.3f; da[2][1] = 0.5f; da[2][2] = -0.2f;	//inside
		oa[3][0] = 5; oa[3][1] = 5; oa[3][2] = 5;  da[3][0] = 1; da[3][1] = 0; da[3][2] = 0;			//misses
		oa[4][0] = -4; oa[4][1] = 0.5f; oa[4][2] = 0.5f;  da[4][0] = 1; da[4][1] = -1e-6f; da[4][2] = 0;	//d+1e-6 == 0
		oa[5][0] = 1.5f; oa[5][1] = 1.5f; oa[5][2] = 4;  da[5][0] = 0; da[5][1] = 0; da[5][2] = -1;		//grazes an edge
	}
	auto [nears, fars] = IntersectWithAABB(o, d, bbox, 0.f);
	save_npy("aabb.bbox", bbox); save_npy("aabb.o", o); save_npy("aabb.d", d);
	save_npy("aabb.near", nears); save_npy("aabb.far", fars);
}

static void g_sample_pdf()
{
	// R8: SamplePDF (Sampler.h:6-43), deterministic (det=true <=> Perturb==0, NeRFRenderer.h:428)
	const int n = 16, nb = 63;
	auto near = synth_tensor({n, 1}, 21u, 0.5f, 2.5f);
	auto far = near + synth_tensor({n, 1}, 22u, 1.0f, 2.5f);
	auto t = torch::linspace(0.f, 1.f, 64, torch::kFloat);
	auto z = near * (1.f - t) + far * t;
	auto bins = .5 * (z.index({"...", Slice(1, None)}) + z.index({"...", Slice(None, -1)}));
	auto w = torch::abs(synth_tensor({n, nb - 1}, 23u, 1.0f));
	{
		auto wa = w.accessor<float, 2>();
		for (int j = 0; j < nb - 1; j++) { wa[0][j] = 0.f; wa[1][j] = 1.f; wa[2][j] = (j == 30) ? 1.f : 0.f; wa[3][j] = (j == 0) ? 5.f : 0.f; wa[4][j] = (j == nb - 2) ? 0.25f : 0.f; }
		for (int j = 0; j < nb - 1; j++) { wa[5][j] = std::exp(-0.5f * (j - 20.f) * (j - 20.f) / 4.f); wa[6][j] = 1e-9f * j; wa[7][j] = (j % 7 == 0) ? 0.3f : 1e-6f; }
	}
	save_npy("sample_pdf.bins", bins.contiguous()); save_npy("sample_pdf.weights", w);
	for (int ns : {128, 192, 5})
	{
		auto s = SamplePDF(bins, w, ns, true);
		save_npy("sample_pdf.samples_" + std::to_string(ns), s);
		// aux_: same ATen ops, same order as Sampler.h:10-13,21-22,28-30 to expose the local index tensors
		auto w2 = w + 1e-8;
		auto pdf = w2 / torch::sum(w2, -1, true);
		auto cdf = torch::cumsum(pdf, -1);
		cdf = torch::cat({torch::zeros_like(cdf.index({"...", Slice(None, 1)})), cdf}, -1);
		auto u = torch::linspace(0.f, 1.f, ns, torch::kFloat).expand({n, ns}).contiguous();
		auto inds = torch::searchsorted(cdf, u, false, true);
		if (ns == 128) { save_npy("sample_pdf.aux_pdf", pdf); save_npy("sample_pdf.aux_cdf", cdf); }
		save_npy("sample_pdf.aux_u_" + std::to_string(ns), u.index({0}));
		save_npy("sample_pdf.aux_inds_" + std::to_string(ns), inds);
	}
	save_npy("sample_pdf.aux_t64", t);
	save_npy("sample_pdf.aux_t192", torch::linspace(0.f, 1.f, 192, torch::kFloat));
}

static void g_pe()
{
	// E1: EmbedderImpl (NeRF.cpp:4-39)
	auto x = synth_tensor({40, 3}, 31u, 1.5f);
	x[0][0] = 0.f; x[0][1] = 1.5f; x[0][2] = -1.5f;
	save_npy("pe.x", x);
	for (int nf : {10, 4, 1 + 1})
	{
		Embedder e("pe", nf);
		auto [out, m] = e->forward(x);
		save_npy("pe.out_" + std::to_string(nf), out);
	}
}

static void g_sh()
{
	// S2: SHEncoderImpl (NeRF.cpp:131-201)
	auto d = synth_tensor({40, 3}, 41u, 1.0f);
	d = d / torch::norm(d, 2, -1, true);
	d[0][0] = 0.f; d[0][1] = 0.f; d[0][2] = 1.f;
	d[1][0] = 1.f; d[1][1] = 0.f; d[1][2] = 0.f;
	save_npy("sh.dirs", d.contiguous());
	for (int deg = 1; deg <= 5; deg++)
	{
		SHEncoder e("sh", 3, deg);
		auto [out, m] = e->forward(d);
		save_npy("sh.out_" + std::to_string(deg), out);
	}
}

static void g_hash()
{
	// H1: HashEmbedderImpl (NeRF.cpp:208-318)
	auto bbox = lego_bbox();
	struct Cfg { const char *tag; int L, F, T, base, fine; int npts; float amp; };
	for (Cfg c : { Cfg{"small", 4, 2, 10, 4, 32, 96, 1e-4f}, Cfg{"f4", 3, 4, 8, 2, 20, 64, 0.5f}, Cfg{"f8", 2, 8, 12, 16, 128, 32, 0.5f},
		Cfg{"full", 16, 2, 19, 16, 512, 256, 0.5f}, Cfg{"full1024", 16, 2, 19, 16, 1024, 64, 1e-4f} })
	{
		std::string tag = std::string("hash_") + c.tag;
		HashEmbedder e("embedder", bbox, c.L, c.F, c.T, c.base, c.fine);
		fill_module(tag, e, 5000u, c.amp, 0.f);
		auto x = synth_tensor({c.npts, 3}, 51u, 1.6f);		//some points fall outside the [-1.5,1.5] box
		{
			auto xa = x.accessor<float, 2>();
			xa[0][0] = 1.5f; xa[0][1] = 1.5f; xa[0][2] = 1.5f;		//max corner: idx == res
			xa[1][0] = -1.5f; xa[1][1] = -1.5f; xa[1][2] = -1.5f;	//min corner
			xa[2][0] = 0.f; xa[2][1] = 0.f; xa[2][2] = 0.f;			//exact lattice point at even resolutions
			xa[3][0] = 1.5f; xa[3][1] = 0.1f; xa[3][2] = -1.5f;
			xa[4][0] = 2.0f; xa[4][1] = 0.1f; xa[4][2] = 0.2f;		//outside -> clamped, mask false
			xa[5][0] = 0.75f; xa[5][1] = -0.75f; xa[5][2] = 0.375f;
		}
		auto [emb, mask] = e->forward(x);
		save_npy(tag + ".cfg", torch::tensor({c.L, c.F, c.T, c.base, c.fine}, torch::kInt32));
		save_npy(tag + ".bbox", bbox);
		save_npy(tag + ".x", x); save_npy(tag + ".emb", emb); save_npy(tag + ".mask", mask);
	}
}

static void g_mlp()
{
	// M2: NeRFSmallImpl (NeRF.cpp:322-412), executor construction NeRFExecutor.h:479-493
	for (int nlc : {4, 3})
	{
		std::string tag = "mlp_small_c" + std::to_string(nlc);
		NeRFSmall m(3, 64, 15, nlc, 64, false, 3, 64, 32, 16, "model");
		fill_module(tag, m, 6000u, 1.6f, 0.f);
		auto x = synth_tensor({96, 48}, 61u, 1.0f);
		auto y = m->forward(x);
		save_npy(tag + ".x", x); save_npy(tag + ".y", y);
	}
	{
		// M2 with the predicted-normals head (use_pred_normal = true, NeRF.cpp:343-347, :393-407; selected by the executor when n_importance == 0, NeRFExecutor.h:487): [p, 7]
		std::string tag = "mlp_small_pn";
		NeRFSmall m(3, 64, 15, 3, 64, true, 3, 64, 32, 16, "model");
		fill_module(tag, m, 6200u, 1.6f, 0.f);
		auto x = synth_tensor({48, 48}, 63u, 1.0f);
		save_npy(tag + ".x", x); save_npy(tag + ".y", m->forward(x));
	}
	{
		// SH degree 8 direction input (main.cpp:188) -> 64-d views
		std::string tag = "mlp_small_v64";
		NeRFSmall m(3, 64, 15, 3, 64, false, 3, 64, 32, 64, "model");
		fill_module(tag, m, 6100u, 1.6f, 0.f);
		auto x = synth_tensor({64, 96}, 62u, 1.0f);
		save_npy(tag + ".x", x); save_npy(tag + ".y", m->forward(x));
	}
	{
		// M1: NeRFImpl (NeRF.cpp:41-126) with view directions
		std::string tag = "mlp_nerf";
		NeRF m(8, 256, 63, 27, 5, std::set<int>{4}, true, "model");
		fill_module(tag, m, 7000u, 1.4f, 0.1f);
		auto x = synth_tensor({64, 90}, 71u, 1.0f);
		save_npy(tag + ".x", x); save_npy(tag + ".y", m->forward(x));
	}
	{
		// M1 without view directions (output_linear branch, NeRF.cpp:121-124)
		std::string tag = "mlp_nerf_noview";
		NeRF m(8, 256, 63, 0, 4, std::set<int>{4}, false, "model");
		fill_module(tag, m, 7100u, 1.4f, 0.1f);
		auto x = synth_tensor({32, 63}, 72u, 1.0f);
		save_npy(tag + ".x", x); save_npy(tag + ".y", m->forward(x));
	}
	{
		// L1: LeRFImpl (LeRF.cpp:28-111), main.cpp:203-213 dims
		std::string tag = "lerf";
		LeRF m(32, 2, 256, 768, 128, "lang_model");
		fill_module(tag, m, 8000u, 1.4f, 0.f);
		auto x = synth_tensor({12, 128}, 81u, 0.5f);
		save_npy(tag + ".x", x); save_npy(tag + ".y", m->forward(x));
	}
}

static void g_truncexp()
{
	// C1 helper: TruncExp forward/backward (CustomOps.cpp:5-15)
	auto x = torch::tensor({-120.f, -100.f, -5.f, -1e-3f, 0.f, 0.5f, 4.9f, 5.f, 5.1f, 20.f}).set_requires_grad(true);
	auto y = torch::autograd::TruncExp::apply(x)[0];
	y.sum().backward();
	save_npy("truncexp.x", x); save_npy("truncexp.y", y); save_npy("truncexp.grad", x.grad());
}

static void g_raw2out()
{
	// C1: RawToOutputs (NeRFRenderer.h:199-282)
	Embedder e("e", 2), ed("ed", 2);
	NeRF m(2, 8, 15, 15, 4, std::set<int>{}, true, "model");
	Spy<Embedder, Embedder, NeRF> spy(e, ed, m);
	for (int S : {64, 192})
	{
		const int n = 12;
		auto raw = synth_tensor({n, S, 4}, 91u + S, 3.0f);
		auto near = synth_tensor({n, 1}, 92u, 0.5f, 2.5f);
		auto far = near + synth_tensor({n, 1}, 93u, 1.0f, 2.5f);
		auto z = near * (1.f - torch::linspace(0.f, 1.f, S, torch::kFloat)) + far * torch::linspace(0.f, 1.f, S, torch::kFloat);
		auto d = synth_tensor({n, 3}, 94u, 1.0f);
		{
			// density scaled so rays span transparent..opaque; ray 0 zero sigma, ray 1 saturated, ray 2 negative sigma
			auto s = raw.index({"...", 3});
			raw.index_put_({"...", 3}, s * 20.f);
			raw.index_put_({0, "...", 3}, 0.f);
			raw.index_put_({1, "...", 3}, 1e4f);
			raw.index_put_({2, "...", 3}, -5.f);
			raw.index_put_({3, Slice(0, S / 2), 3}, 0.f);
			raw.index_put_({3, Slice(S / 2, None), 3}, 50.f);
			z.index_put_({4, Slice()}, z.index({4, 0}).item<float>());		//degenerate ray: all samples at one depth
		}
		std::string tag = "raw2out_" + std::to_string(S);
		save_npy(tag + ".raw", raw); save_npy(tag + ".z", z.contiguous()); save_npy(tag + ".d", d);
		save_outputs(tag + ".black_", spy.OpenRawToOutputs(raw, z, d, false));
		save_outputs(tag + ".white_", spy.OpenRawToOutputs(raw, z, d, true));
	}
}

template <class SpyT>
static void dump_spy(const std::string &tag, SpyT &spy, const NeRFRenderResult &res)
{
	save_npy(tag + ".rays_flat", spy.batch_rays[0]);
	// chunks concatenated; pass 0 = coarse, pass 1 = fine per chunk
	std::vector<torch::Tensor> cp, cr, fp, fr, cz, fz, cw;
	for (size_t i = 0; i + 1 < spy.net_pts.size(); i += 2)
	{
		cp.push_back(spy.net_pts[i]); cr.push_back(spy.net_raw[i]); fp.push_back(spy.net_pts[i + 1]); fr.push_back(spy.net_raw[i + 1]);
		cz.push_back(spy.r2o_z[i]); fz.push_back(spy.r2o_z[i + 1]); cw.push_back(spy.r2o_out[i].Weights);
	}
	save_npy(tag + ".coarse_pts", torch::cat(cp, 0)); save_npy(tag + ".coarse_raw", torch::cat(cr, 0));
	save_npy(tag + ".coarse_z", torch::cat(cz, 0)); save_npy(tag + ".coarse_weights", torch::cat(cw, 0));
	save_npy(tag + ".fine_pts", torch::cat(fp, 0)); save_npy(tag + ".fine_raw", torch::cat(fr, 0));
	save_npy(tag + ".fine_z", torch::cat(fz, 0));
	save_outputs(tag + ".out_", res.Outputs);
	save_npy(tag + ".near_far", torch::tensor({res.Near, res.Far}));
}

// N4: the image-space tail of NeRFExecutor::RenderPath (NeRFExecutor.h:690, :698-700 + TorchTensorToCVMat, NeRFRenderer.h:58-68):
// depth normalisation by the frame's Near/Far, then mul(255).clamp(0,255).to(kU8).  TorchTensorToCVMat itself needs OpenCV (absent);
// its tensor expression is evaluated here verbatim.
static void g_post(const NeRFRenderResult &res)
{
	auto to_u8 = [](torch::Tensor t) { return t.detach().squeeze().cpu().mul(255).clamp(0, 255).to(torch::kU8).contiguous(); };		//NeRFRenderer.h:60-63
	auto depth_n = (res.Outputs.DepthMap - res.Near) / (res.Far - res.Near);		//NeRFExecutor.h:690
	save_npy("post.rgb", res.Outputs.RGBMap); save_npy("post.disp", res.Outputs.DispMap); save_npy("post.depth", res.Outputs.DepthMap);
	save_npy("post.near_far", torch::tensor({res.Near, res.Far}));
	save_npy("post.depth_norm", depth_n);
	save_npy("post.rgb_u8", to_u8(res.Outputs.RGBMap)); save_npy("post.disp_u8", to_u8(res.Outputs.DispMap)); save_npy("post.depth_u8", to_u8(depth_n));
	// edge values: negatives, > 1, exact .5/255 steps
	auto edge = torch::tensor({-0.5f, 0.f, 0.00195f, 0.00392156886f, 0.00392157f, 0.5f, 0.99999994f, 1.f, 1.2f, 300.f, 0.0039215684f * 37.f});
	save_npy("post.edge", edge); save_npy("post.edge_u8", to_u8(edge));
}

static void g_render()
{
	const int h = 8, w = 8;
	auto k = lego_K(h, w);
	auto c2w = orbit_pose(-63.f, -30.f, 4.f);
	// a tighter view so most rays hit the box: scale focal down (wider fov)
	k[0][0] = k[0][0] * 0.8f; k[1][1] = k[1][1] * 0.8f;
	auto bbox = lego_bbox();
	{
		// C3 shape: HashEmbedder(L16,T19,F2,16..512) + SHEncoder(4) + NeRFSmall (NeRFExecutor.h:427-493 construction)
		std::string tag = "render_hash";
		HashEmbedder e("embedder", bbox, 16, 2, 19, 16, 512);
		SHEncoder ed("embeddirs", 3, 4);
		NeRFSmall m(3, 64, 15, 4, 64, false, 3, 64, 32, 16, "model");
		fill_module(tag, e, 5000u, 0.5f, 0.f);
		fill_module(tag, m, 6000u, 1.6f, 0.f, {{"sigma_net_2", 30.0f}});
		save_npy(tag + ".k", k); save_npy(tag + ".c2w", c2w); save_npy(tag + ".bbox", bbox);
		for (int chunk : {64, 24})
		{
			Spy<HashEmbedder, SHEncoder, NeRFSmall> spy(e, ed, m);
			auto res = spy.Render(h, w, k, lego_params(64, 128, chunk), {torch::Tensor(), torch::Tensor(), torch::Tensor()}, c2w);
			if (chunk == 64) { dump_spy(tag, spy, res); g_post(res); }
			else { save_npy(tag + ".chunk24_rgb", res.Outputs.RGBMap); save_npy(tag + ".chunk24_depth", res.Outputs.DepthMap); }
		}
		{
			// LinDisp + black background + explicit ray batch (training-style call, NeRFExecutor.h:876)
			Spy<HashEmbedder, SHEncoder, NeRFSmall> spy(e, ed, m);
			auto [o, d, cone] = GetRays(h, w, k, c2w);
			auto rp = lego_params(64, 128, 64);
			rp.LinDisp = true; rp.WhiteBkgr = false;
			auto res = spy.Render(0, 0, torch::Tensor(), rp, {o.reshape({-1, 3}).index({Slice(0, 40)}), d.reshape({-1, 3}).index({Slice(0, 40)}), cone});
			dump_spy("render_hash_lindisp", spy, res);
		}
		// Stochastic branches (Perturb > 0, cone rays with TangentScatter, training-time noise / preconditioning).  The torch::rand /
		// randn draws the reference makes are replayed after re-seeding (same shapes, same order) and saved next to its outputs, so
		// a restatement fed the SAME draws can be compared value for value.
		for (int variant = 0; variant < 2; variant++)
		{
			std::string stag = variant == 0 ? "render_stoch" : "render_stoch_train";
			const int ns = 32, ni = 48, nr = h * w;
			Spy<HashEmbedder, SHEncoder, NeRFSmall> spy(e, ed, m);
			auto rp = lego_params(ns, ni, nr);
			rp.ThinRay = false; rp.Perturb = 1.f;
			if (variant == 1) { rp.RawNoiseStd = 0.5f; rp.StochasticPreconditioningAlpha = 0.01f; }
			torch::manual_seed(777);
			auto res = spy.Render(h, w, k, rp, {torch::Tensor(), torch::Tensor(), torch::Tensor()}, c2w);
			dump_spy(stag, spy, res);
			save_npy(stag + ".cone_angle", std::get<2>(GetRays(h, w, k, c2w)).reshape({1}));
			torch::manual_seed(777);
			save_npy(stag + ".t_rand", torch::rand({nr, ns}));					//NeRFRenderer.h:415
			save_npy(stag + ".u_r1", torch::rand({nr, ns, 1}));				//:342
			save_npy(stag + ".u_theta1", torch::rand({nr, ns, 1}));		//:343
			if (variant == 1) save_npy(stag + ".noise1", torch::randn({nr, ns}));		//:252
			save_npy(stag + ".u_pdf", torch::rand({nr, ni}));					//Sampler.h:23
			if (variant == 1) save_npy(stag + ".precond", torch::randn({nr, ns + ni, 3}));		//:439
			save_npy(stag + ".u_r2", torch::rand({nr, ns + ni, 1}));
			save_npy(stag + ".u_theta2", torch::rand({nr, ns + ni, 1}));
			if (variant == 1) save_npy(stag + ".noise2", torch::randn({nr, ns + ni}));
		}
		{
			// Ndc + UseViewdirs (NeRFRenderer.h:549-568): the view directions are normalised from the rays of the pose BEFORE NDCRays replaces
			// rays_o / rays_d.  A forward-facing camera (looking down -z, as NDCRays assumes) slightly off the axis; the warped rays run from
			// z = -1 to z = +1 inside the same [-1.5, 1.5]^3 box.
			std::string ntag = "render_ndc";
			auto c2n = torch::tensor({{0.98f, -0.05f, 0.19f, 0.10f}, {0.06f, 0.995f, -0.07f, -0.05f}, {-0.185f, 0.08f, 0.98f, 0.20f}});
			save_npy(ntag + ".k", k); save_npy(ntag + ".c2w", c2n); save_npy(ntag + ".bbox", bbox);
			// Reference quirk: Render() keeps `auto sh = rays_d.sizes()` (:562) -- an ArrayRef INTO the tensor that `std::tie(rays_o, rays_d, ...) = NDCRays(...)`
			// (:567) then releases, so the reshapes at :591-600 read freed memory (here: "shape '[8, 0, 0]' is invalid").  Everything up to and including
			// BatchifyRays has run by then; the spy holds the packed rays and every RawToOutputs result, which is what is saved (per-ray maps, flat).
			auto run = [&](Spy<HashEmbedder, SHEncoder, NeRFSmall> &spy, std::tuple<torch::Tensor, torch::Tensor, torch::Tensor> rays, torch::Tensor pose) {
				auto rp = lego_params(64, 128, 40);
				rp.Ndc = true;
				bool threw = false;
				try { spy.Render(h, w, k, rp, rays, pose); } catch (const c10::Error &) { threw = true; }
				return threw;
			};
			auto cat_fine = [](Spy<HashEmbedder, SHEncoder, NeRFSmall> &spy, const std::string &tag) {
				std::vector<torch::Tensor> rgb, depth, disp, acc, fz, cw;
				for (size_t i = 0; i + 1 < spy.r2o_out.size(); i += 2)
				{
					rgb.push_back(spy.r2o_out[i + 1].RGBMap); depth.push_back(spy.r2o_out[i + 1].DepthMap); disp.push_back(spy.r2o_out[i + 1].DispMap);
					acc.push_back(spy.r2o_out[i + 1].AccMap); fz.push_back(spy.r2o_z[i + 1]); cw.push_back(spy.r2o_out[i].Weights);
				}
				save_npy(tag + "rgb", torch::cat(rgb, 0)); save_npy(tag + "depth", torch::cat(depth, 0)); save_npy(tag + "disp", torch::cat(disp, 0));
				save_npy(tag + "acc", torch::cat(acc, 0)); save_npy(tag + "fine_z", torch::cat(fz, 0)); save_npy(tag + "coarse_weights", torch::cat(cw, 0));
			};
			Spy<HashEmbedder, SHEncoder, NeRFSmall> spy(e, ed, m);
			const bool threw = run(spy, {torch::Tensor(), torch::Tensor(), torch::Tensor()}, c2n);
			save_npy(ntag + ".reference_render_threw", torch::tensor({threw ? 1 : 0}, torch::kInt32));
			save_npy(ntag + ".rays_flat", spy.batch_rays[0]);
			cat_fine(spy, ntag + ".out_");
			save_npy(ntag + ".near_far", torch::stack({spy.batch_rays[0].index({Slice(), 6}).min(), spy.batch_rays[0].index({Slice(), 7}).max()}));		//the expressions of :602-603
			// the same through an explicit ray batch (the first 40 rays of the frame)
			Spy<HashEmbedder, SHEncoder, NeRFSmall> spy2(e, ed, m);
			auto [o, d, cone] = GetRays(h, w, k, c2n);
			run(spy2, {o.reshape({-1, 3}).index({Slice(0, 40)}), d.reshape({-1, 3}).index({Slice(0, 40)}), cone}, torch::Tensor());
			save_npy(ntag + ".batch_rays_flat", spy2.batch_rays[0]);
			cat_fine(spy2, ntag + ".batch_");
		}
		{
			// c2w_staticcam (NeRFRenderer.h:554-558): rays from the static camera, view directions from c2w
			std::string stag = "render_staticcam";
			auto c2s = orbit_pose(-20.f, -35.f, 3.6f);
			Spy<HashEmbedder, SHEncoder, NeRFSmall> spy(e, ed, m);
			auto res = spy.Render(h, w, k, lego_params(64, 128, 64), {torch::Tensor(), torch::Tensor(), torch::Tensor()}, c2w, c2s);
			save_npy(stag + ".k", k); save_npy(stag + ".c2w", c2w); save_npy(stag + ".c2w_staticcam", c2s); save_npy(stag + ".bbox", bbox);
			dump_spy(stag, spy, res);
		}
		{
			// X1: the render-factor step of NeRFExecutor::RenderView (NeRFExecutor.h:618-627).  NeRFExecutor.h is unbuildable here (NeRFactor / RuCLIP
			// headers), so its statements on h, w and k1 are evaluated below with the same types (int / float field, torch::Tensor element division),
			// followed by the reference's own Render -- an "aux_" derivation in the sense of this file's header.
			std::string rtag = "render_factor";
			int hh = 26, ww = 26;
			float RenderFactor = 3;
			auto kk = lego_K(hh, ww);
			kk[0][0] = kk[0][0] * 0.8f; kk[1][1] = kk[1][1] * 0.8f;
			save_npy(rtag + ".k", kk); save_npy(rtag + ".hw", torch::tensor({hh, ww}, torch::kInt32)); save_scalar(rtag + ".render_factor", RenderFactor);
			torch::Tensor k1 = kk.clone().detach();
			hh = hh / RenderFactor;
			ww = ww / RenderFactor;
			k1[0][0] = k1[0][0] / RenderFactor;
			k1[1][1] = k1[1][1] / RenderFactor;
			k1[0][2] = k1[0][2] / RenderFactor;
			k1[1][2] = k1[1][2] / RenderFactor;
			save_npy(rtag + ".aux_k1", k1); save_npy(rtag + ".aux_hw1", torch::tensor({hh, ww}, torch::kInt32)); save_npy(rtag + ".c2w", c2w);
			NeRFRenderer<HashEmbedder, SHEncoder, NeRFSmall> r(e, ed, m);
			auto res = r.Render(hh, ww, k1, lego_params(64, 128, 64), {torch::Tensor(), torch::Tensor(), torch::Tensor()}, c2w);
			save_outputs(rtag + ".out_", res.Outputs);
			save_npy(rtag + ".near_far", torch::tensor({res.Near, res.Far}));
		}
	}
	{
		// C2 shape: PE(10)/PE(4) + NeRF 8x256 with viewdirs
		std::string tag = "render_classic";
		Embedder e("embedder", 10), ed("embeddirs", 4);
		NeRF m(8, 256, 63, 27, 5, std::set<int>{4}, true, "model");
		fill_module(tag, m, 7000u, 1.4f, 0.1f, {{"alpha_linear.weight", 40.0f}});
		save_npy(tag + ".k", k); save_npy(tag + ".c2w", c2w); save_npy(tag + ".bbox", bbox);
		Spy<Embedder, Embedder, NeRF> spy(e, ed, m);
		auto res = spy.Render(h, w, k, lego_params(64, 128, 64), {torch::Tensor(), torch::Tensor(), torch::Tensor()}, c2w);
		dump_spy(tag, spy, res);
	}
	{
		// C1 shape: coarse only, N_importance = 0 -> Render() leaves Outputs undefined (NeRFRenderer.h:423 vs :448);
		// the coarse half is taken from the spy's RawToOutputs record.
		std::string tag = "render_classic_coarse";
		Embedder e("embedder", 10), ed("embeddirs", 4);
		NeRF m(8, 256, 63, 27, 4, std::set<int>{4}, true, "model");
		fill_module(tag, m, 7000u, 1.4f, 0.1f, {{"alpha_linear.weight", 40.0f}});
		Spy<Embedder, Embedder, NeRF> spy(e, ed, m);
		auto res = spy.Render(h, w, k, lego_params(64, 0, 1024), {torch::Tensor(), torch::Tensor(), torch::Tensor()}, c2w);
		save_npy(tag + ".rgb_defined", torch::tensor({res.Outputs.RGBMap.defined() ? 1 : 0}, torch::kInt32));
		save_npy(tag + ".coarse_z", spy.r2o_z[0]); save_npy(tag + ".coarse_raw", spy.r2o_raw[0]);
		save_outputs(tag + ".out_", spy.r2o_out[0]);
	}
}

// ----------------------------------------------------------------------------------------------
// N1: one (and a second) optimisation step of NeRFExecutor::Train (NeRFExecutor.h:862-995) on a tiny HashNeRF:
// Render(ray batch) -> huber_loss(RGBMap, target) -> backward -> Adam(lr, betas (0.9, 0.99), eps 1e-15) (:539).
// Dumps the loss, d loss / d raw of the fine pass, every parameter gradient, and the parameters after each step.
// ----------------------------------------------------------------------------------------------
// The total-variation regulariser of the LibTorch HashEmbedder (NeRF.h:255-300), added to the loss for the first half of training
// (NeRFExecutor.h:896-913): value and gradient w.r.t. the level's table, with the cube's random min vertex replayed.
static void g_tv()
{
	const std::string tag = "tv_loss";
	auto bbox = lego_bbox();
	HashEmbedder e("embedder", bbox, 6, 2, 14, 16, 256);
	fill_module(tag, e, 5000u, 0.5f, 0.f);
	for (int level : {0, 3, 5})
	{
		const std::string st = tag + ".l" + std::to_string(level) + "_";
		for (auto &p : e->parameters()) if (p.grad().defined()) p.grad().zero_();
		torch::manual_seed(4000 + level);
		auto loss = TotalVariationLoss(e, torch::kCPU, 16, 256, level, 14, 6);
		loss.backward();
		double b = exp((log(256.) - log(16.)) / (6 - 1));
		int64_t res = (int64_t)floor(pow(b, level) * 16);
		int64_t cube = std::min<int64_t>(255, std::max<int64_t>(15, (int64_t)floor(res / 10.f)));
		torch::manual_seed(4000 + level);
		auto min_vertex = torch::randint(0, res - cube, {3}, torch::kLong);		//NeRF.h:276
		save_npy(st + "loss", loss.detach().reshape({1}).to(torch::kFloat32));
		save_npy(st + "min_vertex", min_vertex.to(torch::kInt32));
		save_npy(st + "res_cube", torch::tensor({(int)res, (int)cube}, torch::kInt32));
		int li = 0;
		for (auto &p : e->named_parameters()) { if (li == level) save_npy(st + "grad", p.value().grad()); li++; }
	}
}

static void g_train()
{
	const std::string tag = "train_hash";
	const int h = 8, w = 8, ns = 32, ni = 32;
	auto k = lego_K(h, w);
	k[0][0] = k[0][0] * 0.8f; k[1][1] = k[1][1] * 0.8f;
	auto c2w = orbit_pose(-63.f, -30.f, 4.f);
	auto bbox = lego_bbox();
	HashEmbedder e("embedder", bbox, 4, 2, 12, 16, 128);
	SHEncoder ed("embeddirs", 3, 4);
	NeRFSmall m(3, 64, 15, 3, 64, false, 3, 64, 8, 16, "model");
	fill_module(tag, e, 5000u, 0.5f, 0.f);
	fill_module(tag, m, 6000u, 1.6f, 0.f, {{"sigma_net_2", 8.0f}});
	auto [o, d, cone] = GetRays(h, w, k, c2w);
	o = o.reshape({-1, 3}); d = d.reshape({-1, 3});
	auto target = synth_tensor({h * w, 3}, 9100u, 0.5f, 0.5f);
	save_npy(tag + ".rays_o", o); save_npy(tag + ".rays_d", d); save_npy(tag + ".target", target); save_npy(tag + ".bbox", bbox);
	std::vector<torch::Tensor> grad_vars;
	for (auto &p : e->parameters()) grad_vars.push_back(p);
	for (auto &p : m->parameters()) grad_vars.push_back(p);
	const float lr = 5e-3f;
	torch::optim::Adam opt(grad_vars, torch::optim::AdamOptions(lr).eps(1e-15).betas(std::make_tuple(0.9, 0.99)));		//NeRFExecutor.h:539
	save_npy(tag + ".lr", torch::tensor({lr}));
	auto rp = lego_params(ns, ni, h * w);
	rp.WhiteBkgr = false;		//training default
	for (int step = 1; step <= 2; step++)
	{
		const std::string st = tag + ".s" + std::to_string(step) + "_";
		Spy<HashEmbedder, SHEncoder, NeRFSmall> spy(e, ed, m);
		opt.zero_grad();
		auto res = spy.Render(0, 0, torch::Tensor(), rp, {o, d, cone});
		auto mse = torch::mse_loss(res.Outputs.RGBMap, target.detach());
		auto loss = torch::nn::functional::huber_loss(res.Outputs.RGBMap, target.detach());		//:883
		loss.backward();
		save_npy(st + "loss", loss.detach().reshape({1})); save_npy(st + "mse", mse.detach().reshape({1}));
		save_npy(st + "rgb", res.Outputs.RGBMap.detach());
		if (step == 1)
		{
			save_npy(st + "fine_z", spy.r2o_z[1]); save_npy(st + "fine_raw", spy.r2o_raw[1]); save_npy(st + "fine_pts", spy.net_pts[1]);
			save_npy(st + "grad_fine_raw", spy.raw_live[1].grad());
			save_npy(st + "coarse_raw_has_grad", torch::tensor({spy.raw_live[0].grad().defined() ? spy.raw_live[0].grad().abs().max().item<float>() : -1.f}));
			for (auto &p : e->named_parameters()) save_npy(st + "grad_" + p.key(), p.value().grad().defined() ? p.value().grad() : torch::zeros_like(p.value()));
			for (auto &p : m->named_parameters()) save_npy(st + "grad_" + p.key(), p.value().grad().defined() ? p.value().grad() : torch::zeros_like(p.value()));
		}
		opt.step();
		for (auto &p : e->named_parameters()) save_npy(st + "param_" + p.key(), p.value().detach());
		for (auto &p : m->named_parameters()) save_npy(st + "param_" + p.key(), p.value().detach());
	}
}

// ----------------------------------------------------------------------------------------------
// N1, LeRF branch of the optimisation step (NeRFExecutor.h:955-982): LeRFRenderer->Render on the ray batch ->
// huber_loss(RenderedLangEmbedding, target, reduction none, delta 1.25).sum(-1).nanmean() -> lang_loss.backward().
// What carries gradient is the FINE pass (z_samples are detached, LeRFRenderer.cpp:150): LeRFImpl::forward (compiled LeRF.cpp) on the language grid's
// features, the sigma mask (LeRFRenderer.cpp:37-38), RawToLEOutputs' weights, RenderCLIPEmbedding (the reference's inline function, LeRFRenderer.h:45-54).
// LeRFRenderer.cpp itself cannot be built here (it includes RuCLIP's header for Relevancy, which carries no gradient to the loss): its weights are the
// expression of NeRFRenderer::RawToOutputs (NeRFRenderer.h:234-270 == LeRFRenderer.cpp:38-66, read side by side), so the COMPILED RawToOutputs runs on
// [0, 0, 0, sigma_le] rows and its .Weights are used -- no arithmetic is restated.  The language grid is CuHashEmbedder (CUDA-only): its output is a leaf here
// (`emb`), its own backward is pinned by the restatement (oracle/nerf_oracle.c).
// ----------------------------------------------------------------------------------------------
// ----------------------------------------------------------------------------------------------
// N1 for the classic model: NeRFImpl (NeRF.cpp:41-126) is a legal TNeRF of the same train loop (NeRFExecutor.h:862-995).
//   mlp_nerf_bwd*: autograd of the network alone, d sum(y * c) / d (parameters, input), three shapes (view directions + skip; no view directions; the 8 x 256 bench network,
//                  big gradients stored as every 37th element)
//   train_classic: two optimisation steps of the loop body on a small PE(10) / PE(4) network, as train_hash
// ----------------------------------------------------------------------------------------------
static void g_train_classic()
{
	struct Cfg { const char *tag; int d, w, in, views, out; int skip; bool vd; int n; int stride; };
	for (const Cfg &c : {Cfg{"mlp_nerf_bwd", 4, 32, 15, 9, 5, 1, true, 40, 1}, Cfg{"mlp_nerf_bwd_noview", 4, 24, 15, 0, 4, 2, false, 32, 1}, Cfg{"mlp_nerf_bwd_full", 8, 256, 63, 27, 5, 4, true, 24, 37}})
	{
		const std::string tag = c.tag;
		NeRF m(c.d, c.w, c.in, c.views, c.out, std::set<int>{c.skip}, c.vd, "model");
		fill_module(tag, m, 7000u, 1.4f, 0.1f);
		auto x = synth_tensor({c.n, c.in + c.views}, 71u, 1.0f).set_requires_grad(true);
		auto y = m->forward(x);
		auto cw = synth_tensor({c.n, y.size(1)}, 73u, 1.0f);
		(y * cw).sum().backward();
		save_npy(tag + ".dims", torch::tensor({c.d, c.w, c.in, c.views, c.out, c.skip, (int)c.vd, c.stride}, torch::kInt32));
		save_npy(tag + ".x", x); save_npy(tag + ".y", y); save_npy(tag + ".g_out", cw);
		save_npy(tag + ".grad_x", x.grad());
		for (auto &p : m->named_parameters())
		{
			auto g = p.value().grad().defined() ? p.value().grad() : torch::zeros_like(p.value());
			if (c.stride > 1 && g.numel() > 20000) g = g.reshape({-1}).index({Slice(0, None, c.stride)}).contiguous();
			save_npy(tag + ".grad_" + p.key(), g);
		}
	}
	{
		const std::string tag = "train_classic";
		const int h = 8, w = 8, ns = 16, ni = 16;
		auto k = lego_K(h, w);
		k[0][0] = k[0][0] * 0.8f; k[1][1] = k[1][1] * 0.8f;
		auto c2w = orbit_pose(-63.f, -30.f, 4.f);
		Embedder e("embedder", 10), ed("embeddirs", 4);
		NeRF m(4, 64, 63, 27, 5, std::set<int>{1}, true, "model");
		fill_module(tag, m, 7000u, 1.4f, 0.1f, {{"alpha_linear.weight", 12.0f}});
		auto [o, d, cone] = GetRays(h, w, k, c2w);
		o = o.reshape({-1, 3}); d = d.reshape({-1, 3});
		auto target = synth_tensor({h * w, 3}, 9100u, 0.5f, 0.5f);
		save_npy(tag + ".rays_o", o); save_npy(tag + ".rays_d", d); save_npy(tag + ".target", target); save_npy(tag + ".bbox", lego_bbox());
		std::vector<torch::Tensor> grad_vars;
		for (auto &p : m->parameters()) grad_vars.push_back(p);
		const float lr = 5e-3f;
		torch::optim::Adam opt(grad_vars, torch::optim::AdamOptions(lr).eps(1e-15).betas(std::make_tuple(0.9, 0.99)));		//NeRFExecutor.h:539
		save_npy(tag + ".lr", torch::tensor({lr}));
		auto rp = lego_params(ns, ni, h * w);
		rp.WhiteBkgr = false;
		for (int step = 1; step <= 2; step++)
		{
			const std::string st = tag + ".s" + std::to_string(step) + "_";
			Spy<Embedder, Embedder, NeRF> spy(e, ed, m);
			opt.zero_grad();
			auto res = spy.Render(0, 0, torch::Tensor(), rp, {o, d, cone});
			auto mse = torch::mse_loss(res.Outputs.RGBMap, target.detach());
			auto loss = torch::nn::functional::huber_loss(res.Outputs.RGBMap, target.detach());		//:883
			loss.backward();
			save_npy(st + "loss", loss.detach().reshape({1})); save_npy(st + "mse", mse.detach().reshape({1}));
			save_npy(st + "rgb", res.Outputs.RGBMap.detach());
			if (step == 1)
			{
				save_npy(st + "fine_z", spy.r2o_z[1]); save_npy(st + "fine_raw", spy.r2o_raw[1]); save_npy(st + "fine_pts", spy.net_pts[1]);
				save_npy(st + "grad_fine_raw", spy.raw_live[1].grad());
				for (auto &p : m->named_parameters()) save_npy(st + "grad_" + p.key(), p.value().grad().defined() ? p.value().grad() : torch::zeros_like(p.value()));
			}
			opt.step();
			for (auto &p : m->named_parameters()) save_npy(st + "param_" + p.key(), p.value().detach());
		}
	}
}

static void g_train_lerf()
{
	struct Cfg { const char *tag; int geo, layers, hidden, embed, in, n, s; int stride; };
	// `small`: every gradient in full; `main`: main.cpp:203-213 dims, gradients of the big matrices as every `stride`-th element (fixture size)
	for (const Cfg &c : {Cfg{"train_lerf", 8, 2, 32, 48, 16, 6, 24, 1}, Cfg{"train_lerf_main", 32, 2, 256, 768, 128, 4, 12, 61}, Cfg{"train_lerf_l3", 6, 3, 24, 40, 12, 5, 16, 1}})
	{
		const std::string tag = c.tag;
		LeRF m(c.geo, c.layers, c.hidden, c.embed, c.in, "lang_model");
		fill_module(tag, m, 8000u, 1.4f, 0.f);
		const int n = c.n, S = c.s, E = c.embed;
		auto emb = synth_tensor({n * S, c.in}, 83u, 0.5f).set_requires_grad(true);
		auto keep = torch::ones({n * S}, torch::kBool);
		keep.index_put_({3}, false); keep.index_put_({S + 1}, false); keep.index_put_({2 * S + S / 2}, false);
		auto near = synth_tensor({n, 1}, 92u, 0.5f, 2.5f);
		auto far = near + synth_tensor({n, 1}, 93u, 1.0f, 2.5f);
		auto t = torch::linspace(0.f, 1.f, S, torch::kFloat);
		auto z = (near * (1.f - t) + far * t + synth_tensor({n, S}, 95u, 0.02f)).contiguous();
		z = std::get<0>(torch::sort(z, -1));                                  // a fine-pass depth set: sorted, unevenly spaced
		auto d = synth_tensor({n, 3}, 94u, 40.0f);                             // long direction vectors: dists = dz * |d| put sigma_le * dists at O(1), rays span transparent .. opaque
		auto target = torch::nn::functional::normalize(synth_tensor({n, E}, 96u, 1.0f), torch::nn::functional::NormalizeFuncOptions().dim(-1));
		// RunLENetwork (LeRFRenderer.cpp:27-44) downstream of the embedder
		auto outputs_flat = m->forward(emb);
		outputs_flat.index_put_({~keep, -1}, 0);
		auto raw = outputs_flat.view({n, S, E + 1});
		Embedder e0("e", 2), ed0("ed", 2);
		NeRF m0(2, 8, 15, 15, 4, std::set<int>{}, true, "model");
		Spy<Embedder, Embedder, NeRF> spy(e0, ed0, m0);
		auto raw4 = torch::cat({torch::zeros({n, S, 3}), raw.index({"...", Slice(E, E + 1)})}, -1);
		auto weights = spy.OpenRawToOutputs(raw4, z, d, false).Weights;      // == RawToLEOutputs' WeightsLE (LeRFRenderer.cpp:38-66)
		auto le = raw.index({"...", Slice(0, E)});
		auto rendered = RenderCLIPEmbedding(le, weights.unsqueeze(-1));       // LeRFRenderer.h:45-54 (LeRFRenderer.cpp:74)
		auto lang_loss = torch::nn::functional::huber_loss(rendered, target.detach(),
			torch::nn::functional::HuberLossFuncOptions().reduction(torch::kNone).delta(1.25)).sum(-1).nanmean();       // NeRFExecutor.h:970-974
		rendered.retain_grad(); weights.retain_grad();
		lang_loss.backward();
		save_npy(tag + ".emb", emb); save_npy(tag + ".keep", keep); save_npy(tag + ".z", z); save_npy(tag + ".d", d); save_npy(tag + ".target", target);
		save_npy(tag + ".dims", torch::tensor({c.geo, c.layers, c.hidden, c.embed, c.in, n, S, c.stride}, torch::kInt32));
		save_npy(tag + ".weights", weights); save_npy(tag + ".rendered", rendered); save_npy(tag + ".loss", lang_loss.detach().reshape({1}));
		save_npy(tag + ".sigma_le", raw.index({"...", E}).contiguous());
		save_npy(tag + ".grad_rendered", rendered.grad()); save_npy(tag + ".grad_weights", weights.grad());
		save_npy(tag + ".grad_emb", emb.grad());
		for (auto &p : m->named_parameters())
		{
			auto g = p.value().grad().defined() ? p.value().grad() : torch::zeros_like(p.value());
			if (c.stride > 1 && g.numel() > 20000) g = g.reshape({-1}).index({Slice(0, None, c.stride)}).contiguous();
			save_npy(tag + ".grad_" + p.key(), g);
		}
	}
	{
		// what nanmean does with a ray whose target holds a NaN (a pixel without a CLIP embedding): that ray's term leaves the mean ...
		const std::string tag = "train_lerf_nan";
		auto pred = synth_tensor({5, 8}, 97u, 1.0f).set_requires_grad(true);
		auto target = synth_tensor({5, 8}, 98u, 2.0f);
		target.index_put_({2, 3}, std::numeric_limits<float>::quiet_NaN());
		auto loss = torch::nn::functional::huber_loss(pred, target, torch::nn::functional::HuberLossFuncOptions().reduction(torch::kNone).delta(1.25)).sum(-1).nanmean();
		loss.backward();
		save_npy(tag + ".pred", pred); save_npy(tag + ".target", target); save_npy(tag + ".loss", loss.detach().reshape({1})); save_npy(tag + ".grad_pred", pred.grad());
	}
}

// ----------------------------------------------------------------------------------------------
// bench: the reference CPU renderer timed on synthetic Lego-shaped rays (cpu_baseline kind "reference")
// ----------------------------------------------------------------------------------------------
static int run_bench(int argc, const char **argv)
{
	std::string family = argc > 2 ? argv[2] : "hash";
	int h = argc > 3 ? atoi(argv[3]) : 800, w = h;
	int rows = argc > 4 ? atoi(argv[4]) : 4;		//image rows rendered (bounded sample)
	int ns = argc > 5 ? atoi(argv[5]) : 64, ni = argc > 6 ? atoi(argv[6]) : 128;
	int chunk = argc > 7 ? atoi(argv[7]) : 4096;
	int reps = argc > 8 ? atoi(argv[8]) : 1;
	std::ofstream devnull("/dev/null");
	g_manifest.swap(devnull);
	torch::NoGradGuard ng;
	auto k = lego_K(h, w);
	auto c2w = orbit_pose(30.f, -30.f, 4.f);
	auto bbox = lego_bbox();
	auto [o, d, cone] = GetRays(h, w, k, c2w);
	int r0 = h / 2 - rows / 2;
	auto ro = o.index({Slice(r0, r0 + rows)}).reshape({-1, 3}).contiguous();
	auto rd = d.index({Slice(r0, r0 + rows)}).reshape({-1, 3}).contiguous();
	auto rp = lego_params(ns, ni, chunk);
	rp.ReturnRaw = false; rp.ReturnWeights = false;
	double best = 1e30;
	int64_t nrays = ro.size(0);
	auto run = [&](auto &renderer) {
		for (int r = 0; r < reps + 1; r++)
		{
			auto t0 = std::chrono::steady_clock::now();
			auto res = renderer.Render(0, 0, torch::Tensor(), rp, {ro, rd, cone});
			auto t1 = std::chrono::steady_clock::now();
			double s = std::chrono::duration<double>(t1 - t0).count();
			if (r > 0 || reps == 0) best = std::min(best, s);
		}
	};
	if (family == "hash")
	{
		HashEmbedder e("embedder", bbox, 16, 2, 19, 16, 512);
		SHEncoder ed("embeddirs", 3, 4);
		NeRFSmall m(3, 64, 15, 4, 64, false, 3, 64, 32, 16, "model");
		fill_module("b", e, 5000u, 0.5f, 0.f);
		fill_module("b", m, 6000u, 1.6f, 0.f, {{"sigma_net_2", 30.0f}});
		NeRFRenderer<HashEmbedder, SHEncoder, NeRFSmall> r(e, ed, m);
		run(r);
	} else {
		Embedder e("embedder", 10), ed("embeddirs", 4);
		NeRF m(8, 256, 63, 27, 5, std::set<int>{4}, true, "model");
		fill_module("b", m, 7000u, 1.4f, 0.1f, {{"alpha_linear.weight", 40.0f}});
		NeRFRenderer<Embedder, Embedder, NeRF> r(e, ed, m);
		run(r);
	}
	int64_t units = nrays * (ni > 0 ? (ns + ns + ni) : ns);
	printf("{\"family\": \"%s\", \"rays\": %ld, \"units\": %ld, \"seconds\": %.6f, \"units_per_s\": %.1f, \"threads\": %d}\n",
		family.c_str(), (long)nrays, (long)units, best, units / best, at::get_num_threads());
	return 0;
}

// ----------------------------------------------------------------------------------------------
// bench_train: the reference's optimisation step (NeRFExecutor.h:862-995: Render(ray batch) -> huber -> backward -> Adam) on LibTorch CPU
// ----------------------------------------------------------------------------------------------
static int run_bench_train(int argc, const char **argv)
{
	int nrays = argc > 2 ? atoi(argv[2]) : 1024;
	int ns = argc > 3 ? atoi(argv[3]) : 64, ni = argc > 4 ? atoi(argv[4]) : 128;
	int chunk = argc > 5 ? atoi(argv[5]) : 4096;
	int steps = argc > 6 ? atoi(argv[6]) : 1;
	std::ofstream devnull("/dev/null");
	g_manifest.swap(devnull);
	const int h = 800, w = 800;
	auto k = lego_K(h, w);
	auto c2w = orbit_pose(30.f, -30.f, 4.f);
	auto bbox = lego_bbox();
	auto [o, d, cone] = GetRays(h, w, k, c2w);
	auto idx = torch::arange(0, nrays, torch::kLong) * ((int64_t)h * w / nrays);		//a strided sample of the frame's rays
	auto ro = o.reshape({-1, 3}).index_select(0, idx).contiguous(), rd = d.reshape({-1, 3}).index_select(0, idx).contiguous();
	auto target = synth_tensor({nrays, 3}, 9100u, 0.5f, 0.5f);
	HashEmbedder e("embedder", bbox, 16, 2, 19, 16, 512);
	SHEncoder ed("embeddirs", 3, 4);
	NeRFSmall m(3, 64, 15, 4, 64, false, 3, 64, 32, 16, "model");
	fill_module("b", e, 5000u, 0.5f, 0.f);
	fill_module("b", m, 6000u, 1.6f, 0.f, {{"sigma_net_2", 30.0f}});
	NeRFRenderer<HashEmbedder, SHEncoder, NeRFSmall> r(e, ed, m);
	std::vector<torch::Tensor> grad_vars;
	for (auto &p : e->parameters()) grad_vars.push_back(p);
	for (auto &p : m->parameters()) grad_vars.push_back(p);
	torch::optim::Adam opt(grad_vars, torch::optim::AdamOptions(5e-4).eps(1e-15).betas(std::make_tuple(0.9, 0.99)));
	auto rp = lego_params(ns, ni, chunk);
	rp.ReturnRaw = false; rp.ReturnWeights = false; rp.WhiteBkgr = false;
	double best = 1e30; float last_loss = 0.f;
	for (int it = 0; it < steps + 1; it++)
	{
		auto t0 = std::chrono::steady_clock::now();
		opt.zero_grad();
		auto res = r.Render(0, 0, torch::Tensor(), rp, {ro, rd, cone});
		auto loss = torch::nn::functional::huber_loss(res.Outputs.RGBMap, target);
		loss.backward();
		opt.step();
		auto t1 = std::chrono::steady_clock::now();
		double sec = std::chrono::duration<double>(t1 - t0).count();
		if (it > 0 || steps == 0) best = std::min(best, sec);
		last_loss = loss.item<float>();
	}
	int64_t units = (int64_t)nrays * (ns + ns + ni);
	printf("{\"family\": \"hash_train\", \"rays\": %d, \"units\": %ld, \"seconds\": %.6f, \"units_per_s\": %.1f, \"rays_per_s\": %.1f, \"threads\": %d, \"loss\": %.6f}\n",
		nrays, (long)units, best, units / best, nrays / best, at::get_num_threads(), last_loss);
	return 0;
}

// ----------------------------------------------------------------------------------------------
// train_curve: FIFTY optimisation steps of NeRFExecutor::Train's loop body (NeRFExecutor.h:862-996) on the reference's own modules and autograd -- HashEmbedder L16 F2 T2^12
// (16..128) + SHEncoder(4) + NeRFSmall 3x64 / 3x64, torch::optim::Adam(lr 1e-2, betas (0.9, 0.99), eps 1e-15), the learning-rate decay of :992-996 -- fitting a student to
// four 24 x 24 views rendered by a teacher of the same architecture (the reference's own Render, 32 + 32 samples).  Step i trains on 192 pixels of view i % 4:
// pixel index (131 i + 29 j) % 576, j < 192 (a closed form: the test regenerates every batch from GetRays, which is bit-exact against golden `rays`).  Emitted: the teacher
// images, the per-step huber loss and mse, the learning rate in force after every step next to the one :994 computes (they differ: see the loop), the first batch (rays,
// targets) as a check of the batch rule.  The TotalVariationLoss term
// (:896-913) is left out: its cube origins come from torch::randint on the global generator.
// ----------------------------------------------------------------------------------------------
static void g_train_curve()
{
	const std::string tag = "train_curve";
	const int h = 24, w = 24, ns = 32, ni = 32, steps = 50, nrays = 192, nviews = 4;
	auto k = lego_K(h, w);
	auto bbox = lego_bbox();
	const float thetas[nviews] = {-120.f, -30.f, 60.f, 150.f};
	SHEncoder ed("embeddirs", 3, 4);
	auto rp = lego_params(ns, ni, 4096);
	rp.WhiteBkgr = false; rp.ReturnRaw = false;
	// the teacher and its four views
	std::vector<torch::Tensor> img, ro, rd;
	{
		torch::NoGradGuard ng;
		HashEmbedder te("embedder", bbox, 16, 2, 12, 16, 128);
		NeRFSmall tm(3, 64, 15, 3, 64, false, 3, 64, 32, 16, "model");
		fill_module(tag + "_teacher", te, 5000u, 0.5f, 0.f);
		fill_module(tag + "_teacher", tm, 6000u, 1.6f, 0.f, {{"sigma_net_2", 6.0f}});
		NeRFRenderer<HashEmbedder, SHEncoder, NeRFSmall> tr(te, ed, tm);
		for (int v = 0; v < nviews; v++)
		{
			auto c2w = orbit_pose(thetas[v], -30.f, 4.f);
			auto [o, d, cone] = GetRays(h, w, k, c2w);
			auto res = tr.Render(h, w, k, rp, {torch::Tensor(), torch::Tensor(), torch::Tensor()}, c2w);
			img.push_back(res.Outputs.RGBMap.reshape({-1, 3}).contiguous()); ro.push_back(o.reshape({-1, 3}).contiguous()); rd.push_back(d.reshape({-1, 3}).contiguous());
		}
	}
	save_npy(tag + ".teacher_images", torch::stack(img, 0));
	save_npy(tag + ".thetas", torch::from_blob(const_cast<float *>(thetas), {nviews}).clone());
	save_npy(tag + ".dims", torch::tensor({h, w, ns, ni, steps, nrays, nviews}, torch::kInt32));
	save_npy(tag + ".bbox", bbox);
	// the student, trained twice: with LibTorch's default intra-op thread count and with ONE thread.  The two runs start from the same bits and see the same batches; they
	// differ only in the order MKL / at::sum add things up -- the reference's own sensitivity to rounding (Adam with eps 1e-15 turns a rounding-level gradient into a whole
	// step; a fine sample that changes CDF bin moves a pixel), which is the yardstick the HIP Trainer's curve is held to (tests: test_training_loss_curve_*)
	const float LearningRate = 1e-2f, LRateDecay = 0.1f;		//decay_steps = 100
	save_npy(tag + ".lr0_lrate_decay", torch::tensor({LearningRate, LRateDecay}));
	const int default_threads = at::get_num_threads();
	for (int run = 0; run < 2; run++)
	{
		at::set_num_threads(run == 0 ? default_threads : 1);
		std::ofstream devnull("/dev/null");
		if (run == 1) g_manifest.swap(devnull);			//the second student's fill_module lines would repeat the first's
		HashEmbedder e("embedder", bbox, 16, 2, 12, 16, 128);
		NeRFSmall m(3, 64, 15, 3, 64, false, 3, 64, 32, 16, "model");
		fill_module(tag, e, 7770u, 1e-2f, 0.f);
		fill_module(tag, m, 8880u, 1.6f, 0.f);
		if (run == 1) g_manifest.swap(devnull);
		std::vector<torch::Tensor> grad_vars;
		for (auto &p : e->parameters()) grad_vars.push_back(p);
		for (auto &p : m->parameters()) grad_vars.push_back(p);
		torch::optim::Adam opt(grad_vars, torch::optim::AdamOptions(LearningRate).eps(1e-15).betas(std::make_tuple(0.9, 0.99)));		//NeRFExecutor.h:539
		NeRFRenderer<HashEmbedder, SHEncoder, NeRFSmall> renderer(e, ed, m);
		std::vector<float> losses, mses, lrs;
		int global_step = 0;
		for (int i = 0; i < steps; i++)
		{
			const int v = i % nviews;
			std::vector<int64_t> idx(nrays);
			for (int j = 0; j < nrays; j++) idx[j] = (131ll * i + 29ll * j) % (h * w);
			auto it = torch::from_blob(idx.data(), {nrays}, torch::kLong).clone();
			auto o = ro[v].index_select(0, it), d = rd[v].index_select(0, it), target = img[v].index_select(0, it);
			if (i == 0 && run == 0) { save_npy(tag + ".s0_rays_o", o); save_npy(tag + ".s0_rays_d", d); save_npy(tag + ".s0_target", target); }
			opt.zero_grad();																																														//:866
			auto res = renderer.Render(0, 0, torch::Tensor(), rp, {o, d, torch::Tensor()});																						//:876
			auto mse_loss = torch::mse_loss(res.Outputs.RGBMap, target.detach());																									//:882
			auto img_loss = torch::nn::functional::huber_loss(res.Outputs.RGBMap, target.detach());																//:883
			auto loss = img_loss;
			loss.backward();																																																			//:923
			opt.step();																																																						//:985
			const int decay_steps = (int)(LRateDecay * 1000);																																			//:992-996
			const float new_lrate = LearningRate * powf(0.1f, (float)global_step / decay_steps);
			for (auto param_group : opt.param_groups())				//:995-996 VERBATIM, `auto` by value included: OptimizerParamGroup's copy constructor CLONES the options, so set_lr
				param_group.options().set_lr(new_lrate);				//lands on the copy and the optimizer's learning rate never changes -- what the reference does is what the golden records
			global_step++;
			losses.push_back(loss.item<float>()); mses.push_back(mse_loss.item<float>());
			lrs.push_back((float)static_cast<torch::optim::AdamOptions &>(opt.param_groups()[0].options()).lr()); lrs.push_back(new_lrate);
		}
		const std::string sfx = run == 0 ? "" : "_one_thread";
		save_npy(tag + ".loss" + sfx, torch::from_blob(losses.data(), {steps}).clone());
		save_npy(tag + ".mse" + sfx, torch::from_blob(mses.data(), {steps}).clone());
		if (run == 0)
		{
			save_npy(tag + ".lr_in_force_and_lr_computed", torch::from_blob(lrs.data(), {steps, 2}).clone());		//[:, 0] the optimizer's lr after the step, [:, 1] the value :994 computed
			save_npy(tag + ".threads", torch::tensor({default_threads, 1}, torch::kInt32));
		}
	}
	at::set_num_threads(default_threads);
}

// ----------------------------------------------------------------------------------------------
// N3: checkpoint interchange (NeRFExecutor::SaveCheckpoint / LoadCheckpoint, NeRFExecutor.h:540-566, :1055-1070).
//   ckpt_save <dir> : torch::save the train_hash-sized HashEmbedder / NeRFSmall (+ start step) exactly as SaveCheckpoint does, plus a
//                     module carrying CuHashEmbedder's registered names (CuHashEmbedder.cpp:24,73-76; the class itself needs CUDA)
//   ckpt_load <dir> <outdir> : torch::load checkpoints (e.g. written by nerfpp_amd/checkpoint.py) INTO the reference's modules and
//                     dump every parameter as .npy -- proves a file written by this repo restores a reference model
// ----------------------------------------------------------------------------------------------
struct CuHashNamesImpl : torch::nn::Module
{
	torch::Tensor Embeddings, Primes, Biases, FeatLocalSize, FeatLocalIdx;
	CuHashNamesImpl(const std::string &name, int n_levels, int n_feat, int log2_t)
	{
		const int64_t t = 1ll << log2_t;
		Embeddings = register_parameter(name + "_embeddings", torch::zeros({t * n_levels, n_feat}));
		Primes = register_buffer(name + "_primes", torch::zeros({n_levels, 1, 3}, torch::kInt32));
		Biases = register_buffer(name + "_biases", torch::zeros({n_levels, 3}));
		int local_size = (int)((t >> 4) << 4);
		FeatLocalSize = register_buffer(name + "_feat_local_size", torch::full({n_levels}, local_size, torch::kInt32));
		FeatLocalIdx = register_buffer(name + "_feat_local_idx", (torch::cumsum(FeatLocalSize, 0) - local_size).to(torch::kInt32));
	}
};
TORCH_MODULE(CuHashNames);

static int run_ckpt(int argc, const char **argv, bool save)
{
	std::ofstream devnull("/dev/null");
	g_manifest.swap(devnull);
	std::string dir = argv[2];
	auto bbox = lego_bbox();
	HashEmbedder e("embedder", bbox, 4, 2, 12, 16, 128);
	NeRFSmall m(3, 64, 15, 3, 64, false, 3, 64, 8, 16, "model");
	CuHashNames cu("embedder", 4, 2, 12);
	if (save)
	{
		fill_module("ckpt", e, 5000u, 0.5f, 0.f);
		fill_module("ckpt", m, 6000u, 1.6f, 0.f, {{"sigma_net_2", 8.0f}});
		fill_synth(cu->Embeddings, 4242u, 1e-4f);
		{ torch::NoGradGuard ng; auto p = cu->Primes.view({-1}); for (int i = 0; i < p.size(0); i++) p[i] = 268435459 + 2 * i; cu->Biases.fill_(0.25f); }
		torch::save(e, dir + "/embedder_checkpoint.pt");			//NeRFExecutor.h:1058
		torch::save(m, dir + "/model_checkpoint.pt");				//:1059
		torch::save(torch::full({1}, /*value=*/1234), dir + "/start_checkpoint.pt");		//:1066
		torch::save(cu, dir + "/cu_embedder_checkpoint.pt");
		{
			// optimizer_checkpoint.pt (:1067): the reference's Adam (:539) over the embedder's and the model's parameters after one step on a synthetic gradient
			std::vector<torch::Tensor> gv;
			for (auto &p : e->parameters()) gv.push_back(p);
			for (auto &p : m->parameters()) gv.push_back(p);
			torch::optim::Adam opt(gv, torch::optim::AdamOptions(5e-4).eps(1e-15).betas(std::make_tuple(0.9, 0.99)));
			uint32_t k = 0;
			for (auto &p : gv) { p.mutable_grad() = torch::zeros_like(p); fill_synth(p.mutable_grad(), 9000u + 1000u * (k++), 1e-2f); }
			opt.step();
			torch::save(opt, dir + "/optimizer_checkpoint.pt");		//:1067 (a format fixture: the module files above hold the parameters BEFORE this step)
		}
		return 0;
	}
	g_outdir = argv[3];
	torch::load(e, dir + "/embedder_checkpoint.pt");				//:552
	torch::load(m, dir + "/model_checkpoint.pt");					//:553
	torch::load(cu, dir + "/cu_embedder_checkpoint.pt");
	torch::Tensor start; torch::load(start, dir + "/start_checkpoint.pt");
	{
		// NeRFExecutor.h:541-546: the executor restores only when start_checkpoint.pt AND optimizer_checkpoint.pt exist; load the optimizer as it does (:565) and dump its state
		std::vector<torch::Tensor> gv;
		for (auto &p : e->parameters()) gv.push_back(p);
		for (auto &p : m->parameters()) gv.push_back(p);
		torch::optim::Adam opt(gv, torch::optim::AdamOptions(5e-4).eps(1e-15).betas(std::make_tuple(0.9, 0.99)));
		const bool have = std::ifstream(dir + "/optimizer_checkpoint.pt").good();
		save_npy("opt.would_restore", torch::tensor({have ? 1 : 0}, torch::kInt32));
		if (have)
		{
			torch::load(opt, dir + "/optimizer_checkpoint.pt");
			int i = 0;
			for (auto &p : gv)
			{
				auto it = opt.state().find(p.unsafeGetTensorImpl());
				if (it != opt.state().end())
				{
					auto &st = static_cast<torch::optim::AdamParamState &>(*it->second);
					save_npy("opt." + std::to_string(i) + ".exp_avg", st.exp_avg()); save_npy("opt." + std::to_string(i) + ".exp_avg_sq", st.exp_avg_sq());
					save_npy("opt." + std::to_string(i) + ".step", torch::tensor({(float)st.step()}));
				}
				i++;
			}
			save_npy("opt.lr", torch::tensor({(float)static_cast<torch::optim::AdamOptions &>(opt.param_groups()[0].options()).lr()}));
		}
	}
	for (auto &p : e->named_parameters()) save_npy("e." + p.key(), p.value().detach());
	for (auto &p : m->named_parameters()) save_npy("m." + p.key(), p.value().detach());
	for (auto &p : cu->named_parameters()) save_npy("cu." + p.key(), p.value().detach());
	for (auto &p : cu->named_buffers()) save_npy("cu." + p.key(), p.value().detach());
	save_npy("start", start.to(torch::kFloat32));
	return 0;
}

int main(int argc, const char **argv)
{
	if (argc < 2) { std::cerr << "usage: ref_driver golden <outdir> | bench <hash|classic> [h rows ns ni chunk reps]" << std::endl; return 1; }
	std::string cmd = argv[1];
	if (cmd == "bench") return run_bench(argc, argv);
	if (cmd == "bench_train") return run_bench_train(argc, argv);
	if (cmd == "ckpt_save" && argc >= 3) return run_ckpt(argc, argv, true);
	if (cmd == "ckpt_load" && argc >= 4) return run_ckpt(argc, argv, false);
	if (cmd != "golden" || argc < 3) return 1;
	g_outdir = argv[2];
	g_manifest.open(g_outdir + "/manifest.txt");
	torch::manual_seed(42);
	g_rays();
	g_aabb();
	g_sample_pdf();
	g_pe();
	g_sh();
	g_hash();
	g_mlp();
	g_truncexp();
	g_raw2out();
	g_render();
	g_train();
	g_train_classic();
	g_train_lerf();
	g_train_curve();
	g_tv();
	g_manifest.close();
	std::cout << "golden vectors written to " << g_outdir << std::endl;
	return 0;
}

// nerfpp_torch.h -- LibTorch (PyTorch-ROCm C++) adapter: the MI355X path behind the reference's plugin surface.
//
// Header-only glue between the reference's C++/LibTorch host and the C ABI of libnerfpp_hip.so (nerfpp_hip.h).  Tensors are
// used for device memory and the current HIP stream only; every forward is one call into the library.
//
//   HipHashEmbedderImpl  : BaseEmbedderImpl      replaces CuHashEmbedderImpl (CuHashEmbedder.h:8-63) and HashEmbedderImpl
//                                                (NeRF.h:136-209): same ctor (name, bbox, L, F, log2T, base, finest),
//                                                GetOutputDims / forward -> (embedding, keep_mask) / GetBoundingBox /
//                                                Initialize, parameter `<name>_embeddings` + buffers `<name>_primes`,
//                                                `<name>_biases` (CuHashEmbedder.cpp:24,73-76) so checkpoints round-trip.
//   HipSHEncoderImpl     : BaseEmbedderImpl      replaces CuSHEncoderImpl (CuSHEncoder.h:6-29) / SHEncoderImpl (NeRF.h:80-132)
//   HipEmbedderImpl      : BaseEmbedderImpl      replaces EmbedderImpl (NeRF.h:12-31)
//   HipNeRFRenderer<E, D, TNeRF> : NeRFRenderer<E, D, TNeRF>   overrides the virtuals Render / RenderRays / RunNetwork /
//                                                RawToOutputs (NeRFRenderer.h:96-158); BatchifyRays (the chunk loop) is inherited.
//                                                TNeRF is the reference's own NeRFSmall / NeRF module: its parameters are
//                                                read in named_parameters() order (SyncWeights()).
//
// Compile inside the reference tree with -DNRFPP_WITH_REFERENCE (BaseEmbedder.h / NeRF.h / NeRFRenderer.h on the include
// path): the classes then derive from the reference's own bases and slot into NeRFExecutor<...>.
// Without it the header supplies a source-compatible BaseEmbedderImpl so that the encoders can be used on their own.
#pragma once

#include <torch/torch.h>
#include <c10/hip/HIPStream.h>

#include <chrono>
#include <cstdio>
#include <fstream>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "nerfpp_hip.h"

#ifdef NRFPP_WITH_REFERENCE
#include "BaseEmbedder.h"
#include "NeRF.h"
#include "NeRFRenderer.h"
#else
class BaseEmbedderImpl : public torch::nn::Module {
public:
	BaseEmbedderImpl(const std::string &module_name) : torch::nn::Module(module_name) {}
	virtual ~BaseEmbedderImpl() {}
	virtual int GetOutputDims() { return 0; }
	virtual std::pair<torch::Tensor, torch::Tensor> forward(torch::Tensor x) { return std::make_pair(torch::Tensor(), torch::Tensor()); }
};
#endif

namespace nrfpp {

inline void check(int status, const char *what)
{
	if (status != NRF_OK)
		throw std::runtime_error(std::string(what) + ": " + nrf_status_string(status) + ": " + nrf_last_error());   // the reference surfaces errors as c10::Error / exceptions
}

inline void *current_stream() { return (void *)c10::hip::getCurrentHIPStream().stream(); }

inline torch::Tensor dev_f32(torch::Tensor t)
{
	TORCH_CHECK(t.is_cuda(), "nerfpp_torch: tensors must live on the GPU (no CPU fallback)");
	return t.to(torch::kFloat32).contiguous();
}

inline std::vector<float> host_floats(torch::Tensor t) { auto c = t.detach().to(torch::kCPU, torch::kFloat32).contiguous(); return std::vector<float>(c.data_ptr<float>(), c.data_ptr<float>() + c.numel()); }

// ---------------------------------------------------------------------------------------------------------------------
// position / direction encoders
// ---------------------------------------------------------------------------------------------------------------------
class HipEmbedderImpl : public BaseEmbedderImpl {
	int Multires;
public:
	HipEmbedderImpl(const std::string &module_name, int multires) : BaseEmbedderImpl(module_name), Multires(multires) {}
	int GetMultires() const { return Multires; }
	int GetOutputDims() override { return 3 + 6 * Multires; }
	std::pair<torch::Tensor, torch::Tensor> forward(torch::Tensor x) override
	{
		x = dev_f32(x).view({-1, 3});
		auto out = torch::empty({x.size(0), GetOutputDims()}, x.options());
		check(nrf_pe_encode(x.data_ptr<float>(), x.size(0), Multires, out.data_ptr<float>(), current_stream()), "nrf_pe_encode");
		return std::make_pair(out, torch::Tensor());
	}
};
TORCH_MODULE(HipEmbedder);

class HipSHEncoderImpl : public BaseEmbedderImpl {
	int Degree, Variant;
public:
	/// variant: NRF_SH_CUDA reproduces CuSHEncoder (degree <= 8), NRF_SH_LIBTORCH reproduces SHEncoder (degree <= 5)
	HipSHEncoderImpl(const std::string &module_name, const int input_dim = 3, const int degree = 4, const int variant = NRF_SH_CUDA)
		: BaseEmbedderImpl(module_name), Degree(degree), Variant(variant) { TORCH_CHECK(input_dim == 3); }
	int GetDegree() const { return Degree; }
	int GetVariant() const { return Variant; }
	int GetOutputDims() override { return Degree * Degree; }
	std::pair<torch::Tensor, torch::Tensor> forward(torch::Tensor input) override
	{
		input = dev_f32(input).view({-1, 3});
		auto out = torch::empty({input.size(0), GetOutputDims()}, input.options());
		check(nrf_sh_encode(input.data_ptr<float>(), input.size(0), Degree, Variant, out.data_ptr<float>(), current_stream()), "nrf_sh_encode");
		return std::make_pair(out, torch::Tensor());
	}
};
TORCH_MODULE(HipSHEncoder);

class HipHashEmbedderImpl : public BaseEmbedderImpl {
	nrf_hash *Handle = nullptr;
public:
	torch::Tensor BoundingBox;
	int NLevels, NFeaturesPerLevel, Log2HashmapSize, BaseResolution, FinestResolution, Mode;
	torch::Tensor Embeddings, Primes, Biases;

	/// mode NRF_HASH_CU: CuHashEmbedder semantics (fp16 table, per-level primes);  NRF_HASH_NGP: HashEmbedder semantics.
	HipHashEmbedderImpl(const std::string &module_name, torch::Tensor bounding_box, const int n_levels = 16, const int n_features_per_level = 2,
		const int log2_hashmap_size = 19, const int base_resolution = 16, const int finest_resolution = 512, const int mode = NRF_HASH_CU)
		: BaseEmbedderImpl(module_name), BoundingBox(bounding_box), NLevels(n_levels), NFeaturesPerLevel(n_features_per_level),
		Log2HashmapSize(log2_hashmap_size), BaseResolution(base_resolution), FinestResolution(finest_resolution), Mode(mode)
	{
		nrf_hash_desc d{mode, n_levels, n_features_per_level, log2_hashmap_size, base_resolution, finest_resolution, {0, 0, 0, 0, 0, 0}};
		auto bb = host_floats(bounding_box);
		TORCH_CHECK(bb.size() == 6, "bounding_box must hold [min xyz, max xyz]");
		for (int i = 0; i < 6; i++) d.bbox[i] = bb[i];
		check(nrf_hash_create(&d, &Handle), "nrf_hash_create");
		const int64_t rows = ((int64_t)1 << log2_hashmap_size) * n_levels;
		// same parameter / buffer names and init as the reference (CuHashEmbedder.cpp:24: U(0,1)*1e-4; NeRF.cpp:270: U(-1e-4,1e-4))
		auto init = (mode == NRF_HASH_CU) ? torch::rand({rows, n_features_per_level}) * 1e-4f : (torch::rand({rows, n_features_per_level}) * 2.f - 1.f) * 1e-4f;
		Embeddings = register_parameter(module_name + "_embeddings", init.to(torch::kCUDA), /*requires_grad=*/true);
		Primes = register_buffer(module_name + "_primes", torch::zeros({n_levels, 1, 3}, torch::kInt32));
		Biases = register_buffer(module_name + "_biases", torch::zeros({n_levels, 3}, torch::kFloat32));
	}
	~HipHashEmbedderImpl() override { nrf_hash_destroy(Handle); }

	const nrf_hash *GetHandle() const { return Handle; }
	torch::Tensor GetBoundingBox() const { return BoundingBox; }
	int GetOutputDims() override { return NLevels * NFeaturesPerLevel; }

	/// The reference calls Initialize() after construction / checkpoint load (NeRFExecutor.h:570): push table + primes to the library.
	void Initialize() { Sync(); }
	void SetPrimes(torch::Tensor primes) { Primes.copy_(primes.view_as(Primes)); }
	void Sync()
	{
		auto emb = dev_f32(Embeddings.detach());
		check(nrf_hash_set_table(Handle, emb.data_ptr<float>(), 1, current_stream()), "nrf_hash_set_table");
		if (Mode == NRF_HASH_CU) {
			auto p = Primes.to(torch::kCPU, torch::kInt32).contiguous();
			auto b = host_floats(Biases);
			check(nrf_hash_set_primes(Handle, p.data_ptr<int32_t>(), b.data()), "nrf_hash_set_primes");
		}
		c10::hip::getCurrentHIPStream().synchronize();
	}
	std::pair<torch::Tensor, torch::Tensor> forward(torch::Tensor x) override
	{
		x = dev_f32(x).view({-1, 3});
		auto out = torch::empty({x.size(0), GetOutputDims()}, x.options());
		auto mask = torch::empty({x.size(0)}, x.options().dtype(torch::kUInt8));
		check(nrf_hash_encode(Handle, x.data_ptr<float>(), x.size(0), out.data_ptr<float>(), mask.data_ptr<uint8_t>(), current_stream()), "nrf_hash_encode");
		return std::make_pair(out, mask.to(torch::kBool));
	}
};
TORCH_MODULE(HipHashEmbedder);

// ---------------------------------------------------------------------------------------------------------------------
// Multi-GPU: one process per GPU, frames partitioned into contiguous row tiles (nrf_tile_partition), one RCCL all-gather per
// frame over xGMI (nrf_allgather_tiles).  No reference counterpart (the reference is single-GPU).
// ---------------------------------------------------------------------------------------------------------------------
class TileComm {
	nrf_comm *Comm = nullptr;
	int World = 1, Rank = 0;
public:
	/// Rendezvous through the filesystem (launchers that give every rank WORLD_SIZE / RANK and a shared directory need nothing else): rank 0 asks the
	/// library for the RCCL unique id and publishes it at `id_path` (written to a temporary name, then renamed: readers never see a partial file); the other
	/// ranks wait for it.  Call after hipSetDevice / c10::hip::set_device: the communicator binds to the calling thread's current device.
	TileComm(int world, int rank, const std::string &id_path, double timeout_s = 120.0) : World(world), Rank(rank)
	{
		unsigned char id[NRF_COMM_ID_BYTES];
		if (rank == 0) {
			check(nrf_comm_unique_id(id), "nrf_comm_unique_id");
			if (world > 1) {
				const std::string tmp = id_path + ".tmp";
				{ std::ofstream f(tmp, std::ios::binary); f.write(reinterpret_cast<const char *>(id), sizeof(id)); }
				if (std::rename(tmp.c_str(), id_path.c_str()) != 0) throw std::runtime_error("TileComm: cannot publish the communicator id at " + id_path);
			}
		} else {
			const auto t0 = std::chrono::steady_clock::now();
			for (;;) {
				std::ifstream f(id_path, std::ios::binary);
				if (f && f.read(reinterpret_cast<char *>(id), sizeof(id)) && f.gcount() == (std::streamsize)sizeof(id)) break;
				if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) throw std::runtime_error("TileComm: no communicator id at " + id_path);
				std::this_thread::sleep_for(std::chrono::milliseconds(20));
			}
		}
		check(nrf_comm_create(id, world, rank, &Comm), "nrf_comm_create");
	}
	/// Adopt a live ncclComm_t the host already owns (not destroyed here).
	explicit TileComm(void *nccl_comm) { check(nrf_comm_wrap(nccl_comm, &Comm), "nrf_comm_wrap"); World = nrf_comm_world(Comm); Rank = nrf_comm_rank(Comm); }
	TileComm(const TileComm &) = delete;
	TileComm &operator=(const TileComm &) = delete;
	~TileComm() { nrf_comm_destroy(Comm); }

	int GetWorld() const { return World; }
	int GetRank() const { return Rank; }
	/// rows [row0, row0 + rows) of an h-row frame belong to this rank
	std::pair<int, int> Rows(int h) const { int row0 = 0, rows = 0; check(nrf_tile_partition(h, World, Rank, &row0, &rows), "nrf_tile_partition"); return {row0, rows}; }
	/// tiles [F, rows_rank, W, C] (this rank's rows of F frames) -> [F, h, W, C] on every rank; one fused RCCL launch on the current stream
	torch::Tensor AllGatherFrames(torch::Tensor tiles, int h) const
	{
		tiles = dev_f32(tiles);
		TORCH_CHECK(tiles.dim() == 4 && tiles.size(1) == Rows(h).second, "AllGatherFrames: tiles must be [frames, rows of this rank, w, c]");
		auto out = torch::empty({tiles.size(0), (int64_t)h, tiles.size(2), tiles.size(3)}, tiles.options());
		check(nrf_allgather_tiles(Comm, tiles.data_ptr<float>(), (int)tiles.size(0), h, (int)tiles.size(2), (int)tiles.size(3), out.data_ptr<float>(), current_stream()), "nrf_allgather_tiles");
		return out;
	}
};

// ---------------------------------------------------------------------------------------------------------------------
// MLP handle built from a reference module's parameters (named_parameters() order == the blob order of nerfpp_hip.h)
// ---------------------------------------------------------------------------------------------------------------------
template <class TModule>
inline std::vector<float> parameter_blob(TModule &module)
{
	std::vector<float> blob;
	for (auto &p : module->named_parameters()) {
		auto v = host_floats(p.value());
		blob.insert(blob.end(), v.begin(), v.end());
	}
	return blob;
}

struct MlpHandle {
	nrf_mlp *m = nullptr;
	~MlpHandle() { nrf_mlp_destroy(m); }
	void reset(nrf_mlp *n) { nrf_mlp_destroy(m); m = n; }
};

#ifdef NRFPP_WITH_REFERENCE
// ---------------------------------------------------------------------------------------------------------------------
// The renderer: NeRFRenderer<TEmbedder, TEmbedDirs, TNeRF> with its virtuals routed to the HIP path.
// ---------------------------------------------------------------------------------------------------------------------
template <class TEmbedder, class TEmbedDirs, class TNeRF>
class HipNeRFRenderer : public NeRFRenderer<TEmbedder, TEmbedDirs, TNeRF> {
	using Base = NeRFRenderer<TEmbedder, TEmbedDirs, TNeRF>;
	MlpHandle Mlp;
	nrf_renderer *Renderer = nullptr;
	torch::Tensor Workspace;
	int Precision;
	uint64_t Seed = 0;        ///seed of the counter-based draws of the stochastic branches (include/nrf_rng.h)
	int64_t RayCursor = 0;    ///rays already rendered by the current Render() call: makes the draws independent of Chunk

	void *workspace(size_t bytes, torch::Device dev)
	{
		if (!Workspace.defined() || (size_t)Workspace.numel() < bytes) Workspace = torch::empty({(int64_t)bytes}, torch::TensorOptions().dtype(torch::kUInt8).device(dev));
		return Workspace.data_ptr();
	}
public:
	/// small = {num_layers, hidden_dim, geo_feat_dim, num_layers_color, hidden_dim_color} of the NeRFSmall the executor built
	/// (NeRFExecutor.h:479-493), or nerf = {depth, width, output_ch, skip, use_viewdirs} for the classic NeRF.
	HipNeRFRenderer(TEmbedder embed_fn, TEmbedDirs embeddirs_fn, TNeRF nerf, int precision = NRF_PREC_F16_MFMA) : Base(embed_fn, embeddirs_fn, nerf), Precision(precision) {}
	~HipNeRFRenderer() override { nrf_renderer_destroy(Renderer); }

	void SetPrecision(int precision) { Precision = precision; }
	void SetSeed(uint64_t seed) { Seed = seed; }

	/// (Re)read the network's parameters and rebuild the device-side images; call after construction, load or an optimizer step.
	void SyncWeights(const nrf_mlp_small_desc *small, const nrf_mlp_nerf_desc *classic)
	{
		auto blob = parameter_blob(this->NeRF);
		nrf_mlp *m = nullptr;
		if (small) { TORCH_CHECK((int64_t)blob.size() == nrf_mlp_small_param_count(small), "NeRFSmall parameter count mismatch"); check(nrf_mlp_small_create(small, blob.data(), 0, current_stream(), &m), "nrf_mlp_small_create"); }
		else { TORCH_CHECK((int64_t)blob.size() == nrf_mlp_nerf_param_count(classic), "NeRF parameter count mismatch"); check(nrf_mlp_nerf_create(classic, blob.data(), 0, current_stream(), &m), "nrf_mlp_nerf_create"); }
		Mlp.reset(m);
		nrf_renderer_destroy(Renderer); Renderer = nullptr;
		nrf_renderer_desc d{};
		if constexpr (std::is_same_v<TEmbedder, HipHashEmbedder>) { this->EmbedFn->Sync(); d.hash = this->EmbedFn->GetHandle(); }
		else { d.hash = nullptr; d.pe_freqs = this->EmbedFn->GetMultires(); }
		if constexpr (std::is_same_v<TEmbedDirs, HipSHEncoder>) { d.dirs_encoder = this->EmbeddirsFn->GetVariant() == NRF_SH_CUDA ? NRF_DIRS_SH_CUDA : NRF_DIRS_SH_LIBTORCH; d.dirs_param = this->EmbeddirsFn->GetDegree(); }
		else { d.dirs_encoder = NRF_DIRS_PE; d.dirs_param = this->EmbeddirsFn->GetMultires(); }
		d.mlp = Mlp.m;
		check(nrf_renderer_create(&d, &Renderer), "nrf_renderer_create");
	}

protected:
	/// NeRFRenderer.h:164-194
	torch::Tensor RunNetwork(torch::Tensor inputs, torch::Tensor view_dirs, TNeRF fn, TEmbedder embed_fn, TEmbedDirs embeddirs_fn) override
	{
		auto pts = dev_f32(inputs);
		const int64_t n = pts.size(0); const int s = (int)pts.size(1);
		torch::Tensor vd; if (view_dirs.defined() && view_dirs.numel()) vd = dev_f32(view_dirs);
		auto raw = torch::empty({n, s, 4}, pts.options());
		const size_t wsb = nrf_run_network_workspace_bytes(Renderer, n, s);
		check(nrf_run_network(Renderer, pts.data_ptr<float>(), vd.defined() ? vd.data_ptr<float>() : nullptr, n, s, NRF_PREC_F32, raw.data_ptr<float>(),
			workspace(wsb, pts.device()), wsb, current_stream()), "nrf_run_network");
		return raw;
	}

	/// NeRFRenderer.h:199-282
	NeRFRendererOutputs RawToOutputs(torch::Tensor raw, torch::Tensor cone_angle, torch::Tensor z_vals, torch::Tensor rays_d,
		const float raw_noise_std = 0.f, const bool white_bkgr = false) override
	{
		TORCH_CHECK(raw_noise_std == 0.f, "RawToOutputs: raw_noise_std > 0 draws inside RenderRays (nrf_render_rays) or takes explicit draws (nrf_raw2outputs_noise)");
		raw = dev_f32(raw); z_vals = dev_f32(z_vals); rays_d = dev_f32(rays_d);
		const int64_t n = raw.size(0); const int s = (int)raw.size(1), c = (int)raw.size(2);
		NeRFRendererOutputs o;
		o.RGBMap = torch::empty({n, 3}, raw.options()); o.DispMap = torch::empty({n}, raw.options()); o.AccMap = torch::empty({n}, raw.options());
		o.Weights = torch::empty({n, s}, raw.options()); o.DepthMap = torch::empty({n}, raw.options());
		check(nrf_raw2outputs(raw.data_ptr<float>(), z_vals.data_ptr<float>(), rays_d.data_ptr<float>(), 3, n, s, c, white_bkgr, o.RGBMap.data_ptr<float>(),
			o.DispMap.data_ptr<float>(), o.AccMap.data_ptr<float>(), o.Weights.data_ptr<float>(), o.DepthMap.data_ptr<float>(), current_stream()), "nrf_raw2outputs");
		return o;
	}

public:
	/// NeRFRenderer.h:530-605.  Ray generation, view-direction normalisation, AABB clipping and the ray-batch assembly run in
	/// the library (bit-identical to the LibTorch CPU path); the chunk loop is the reference's own BatchifyRays.  NDC scenes and
	/// c2w_staticcam take the inherited torch-op path.
	NeRFRenderResult Render(const int h, const int w, torch::Tensor k, const NeRFRenderParams &render_params,
		std::tuple<torch::Tensor, torch::Tensor, torch::Tensor> rays = {torch::Tensor(), torch::Tensor(), torch::Tensor()},
		torch::Tensor c2w = torch::Tensor(), torch::Tensor c2w_staticcam = torch::Tensor()) override
	{
		if (render_params.Ndc || (c2w_staticcam.defined() && c2w_staticcam.numel() != 0)) return Base::Render(h, w, k, render_params, rays, c2w, c2w_staticcam);
		return RenderRows(h, w, k, render_params, rays, c2w, 0, h);
	}

	/// Rows [row0, row0 + rows) of the h x w frame seen from c2w: one rank's share of a frame (multi-GPU row tiles).  Ray r of the tile is pixel
	/// (row0 + r / w, r % w); the counter-based draws of the stochastic branches are keyed by the ray's position in the WHOLE frame, so a tile equals
	/// the corresponding slice of the full render bit for bit.  Outputs are [rows, w, ...]; Near / Far are the tile's.
	NeRFRenderResult RenderTile(const int h, const int w, torch::Tensor k, const NeRFRenderParams &render_params, torch::Tensor c2w, const int row0, const int rows)
	{
		TORCH_CHECK(!render_params.Ndc, "RenderTile: NDC scenes are rendered whole (Render)");
		TORCH_CHECK(row0 >= 0 && rows >= 0 && row0 + rows <= h, "RenderTile: rows outside the frame");
		return RenderRows(h, w, k, render_params, {torch::Tensor(), torch::Tensor(), torch::Tensor()}, c2w, row0, rows);
	}

	/// One frame over all ranks of `comm`: this rank renders its row tile, ONE all-gather (rgb, disparity, accumulation, depth packed per pixel) returns the
	/// whole frame to every rank.  Near / Far are the whole frame's (rays are cheap: every rank generates the full ray batch for that reduction).
	NeRFRenderResult RenderSharded(const int h, const int w, torch::Tensor k, const NeRFRenderParams &render_params, torch::Tensor c2w, const TileComm &comm)
	{
		const auto [row0, rows] = comm.Rows(h);
		NeRFRenderParams rp = render_params;
		rp.ReturnRaw = false; rp.ReturnWeights = false;                 // per-sample tensors stay on the rank that made them
		NeRFRenderResult tile = RenderTile(h, w, k, rp, c2w, row0, rows);
		auto &o = tile.Outputs;
		auto packed = torch::cat({o.RGBMap.reshape({rows, w, 3}), o.DispMap.reshape({rows, w, 1}), o.AccMap.reshape({rows, w, 1}), o.DepthMap.reshape({rows, w, 1})}, -1).unsqueeze(0);
		auto frame = comm.AllGatherFrames(packed, h).squeeze(0);        // [h, w, 6]
		using torch::indexing::Slice;
		NeRFRenderResult res;
		res.Outputs.RGBMap = frame.index({Slice(), Slice(), Slice(0, 3)}).contiguous();
		res.Outputs.DispMap = frame.index({Slice(), Slice(), 3}).contiguous();
		res.Outputs.AccMap = frame.index({Slice(), Slice(), 4}).contiguous().reshape({-1});
		res.Outputs.DepthMap = frame.index({Slice(), Slice(), 5}).contiguous();
		// Near / Far of the whole frame (NeRFRenderer.h:602-603)
		const auto dev = torch::Device(torch::kCUDA, c10::hip::getCurrentHIPStream().device_index());
		auto K = host_floats(k), M = host_floats(c2w.index({Slice(torch::indexing::None, 3), Slice(torch::indexing::None, 4)}));
		auto ro = torch::empty({(int64_t)h * w, 3}, torch::TensorOptions().dtype(torch::kFloat32).device(dev)), rd = torch::empty_like(ro);
		check(nrf_get_rays(h, w, K.data(), M.data(), 0, h, ro.data_ptr<float>(), rd.data_ptr<float>(), nullptr, current_stream()), "nrf_get_rays");
		auto bb = host_floats(render_params.BoundingBox);
		auto rays_ = torch::empty({(int64_t)h * w, 8}, ro.options());
		check(nrf_pack_rays(ro.data_ptr<float>(), rd.data_ptr<float>(), bb.data(), (int64_t)h * w, 0, rays_.data_ptr<float>(), current_stream()), "nrf_pack_rays");
		check(nrf_near_far_range(rays_.data_ptr<float>(), (int64_t)h * w, 8, &res.Near, &res.Far, current_stream()), "nrf_near_far_range");
		return res;
	}

private:
	NeRFRenderResult RenderRows(const int h, const int w, torch::Tensor k, const NeRFRenderParams &render_params,
		std::tuple<torch::Tensor, torch::Tensor, torch::Tensor> rays, torch::Tensor c2w, const int row0, const int rows)
	{
		const auto dev = torch::Device(torch::kCUDA, c10::hip::getCurrentHIPStream().device_index());
		torch::Tensor rays_o, rays_d, cone_angle;
		const bool from_pose = c2w.defined() && c2w.numel() != 0;
		if (from_pose) {
			auto K = host_floats(k), M = host_floats(c2w.index({torch::indexing::Slice(torch::indexing::None, 3), torch::indexing::Slice(torch::indexing::None, 4)}));
			rays_o = torch::empty({rows, w, 3}, torch::TensorOptions().dtype(torch::kFloat32).device(dev)); rays_d = torch::empty_like(rays_o);
			float cone = 0.f;
			check(nrf_get_rays(h, w, K.data(), M.data(), row0, rows, rays_o.data_ptr<float>(), rays_d.data_ptr<float>(), &cone, current_stream()), "nrf_get_rays");
			cone_angle = torch::tensor(cone);
		} else {
			std::tie(rays_o, rays_d, cone_angle) = rays;
			rays_o = dev_f32(rays_o); rays_d = dev_f32(rays_d);
		}
		auto sh = rays_d.sizes().vec();
		auto o = rays_o.reshape({-1, 3}).contiguous(), d = rays_d.reshape({-1, 3}).contiguous();
		const int64_t n = o.size(0);
		const int stride = render_params.UseViewdirs ? 11 : 8;
		auto bb = host_floats(render_params.BoundingBox);
		auto rays_ = torch::empty({n, stride}, o.options());
		check(nrf_pack_rays(o.data_ptr<float>(), d.data_ptr<float>(), bb.data(), n, render_params.UseViewdirs, rays_.data_ptr<float>(), current_stream()), "nrf_pack_rays");
		RayCursor = from_pose ? (int64_t)row0 * w : 0;
		NeRFRenderResult all_ret = this->BatchifyRays(rays_, render_params.ThinRay ? torch::Tensor() : cone_angle, render_params.NSamples, render_params.Chunk,
			render_params.ReturnRaw, render_params.LinDisp, render_params.Perturb, render_params.NImportance, render_params.WhiteBkgr, render_params.RawNoiseStd,
			render_params.StochasticPreconditioningAlpha, render_params.BoundingBox, render_params.ReturnWeights);
		if (all_ret.Outputs.RGBMap.defined() && all_ret.Outputs.RGBMap.numel() != 0) all_ret.Outputs.RGBMap = torch::reshape(all_ret.Outputs.RGBMap, sh);
		if (sh.size() > 2) {
			if (all_ret.Outputs.DispMap.defined() && all_ret.Outputs.DispMap.numel() != 0) all_ret.Outputs.DispMap = torch::reshape(all_ret.Outputs.DispMap, {sh[0], sh[1]});
			if (all_ret.Outputs.DepthMap.defined() && all_ret.Outputs.DepthMap.numel() != 0) all_ret.Outputs.DepthMap = torch::reshape(all_ret.Outputs.DepthMap, {sh[0], sh[1]});
		}
		check(nrf_near_far_range(rays_.data_ptr<float>(), n, stride, &all_ret.Near, &all_ret.Far, current_stream()), "nrf_near_far_range");
		return all_ret;
	}

public:
	/// NeRFRenderer.h:366-459, one fused call per chunk of packed rays
	NeRFRenderResult RenderRays(torch::Tensor ray_batch, torch::Tensor cone_angle, const int n_samples, const bool return_raw = false,
		const bool lin_disp = false, const float perturb = 0.f, const int n_importance = 0, const bool white_bkgr = false,
		const float raw_noise_std = 0.f, const float stochastic_preconditioning_alpha = 0.f, torch::Tensor bounding_box = torch::Tensor(),
		const bool return_weights = true) override
	{
		auto rays = dev_f32(ray_batch);
		const int64_t n = rays.size(0); const int stride = (int)rays.size(1);
		const int sf = n_samples + n_importance, so = n_importance > 0 ? sf : n_samples;
		auto opt = rays.options();
		auto t = torch::linspace(0.f, 1.f, n_samples, torch::kFloat).to(rays.device());                       // NeRFRenderer.h:393
		torch::Tensor u; if (n_importance > 0) u = torch::linspace(0.f, 1.f, n_importance, torch::kFloat).to(rays.device());   // Sampler.h:21
		NeRFRenderResult res;
		res.Outputs.RGBMap = torch::empty({n, 3}, opt); res.Outputs.DispMap = torch::empty({n}, opt); res.Outputs.AccMap = torch::empty({n}, opt);
		res.Outputs.DepthMap = torch::empty({n}, opt);
		if (return_weights) res.Outputs.Weights = torch::empty({n, so}, opt);
		if (return_raw) res.Raw = torch::empty({n, so, 4}, opt);
		nrf_render_params p{n_samples, n_importance, lin_disp, white_bkgr, Precision, 8};
		// stochastic branches: the library draws from its counter RNG keyed by (Seed, position of the ray in this Render call, sample)
		p.perturb = perturb; p.raw_noise_std = raw_noise_std; p.precond_alpha = stochastic_preconditioning_alpha;
		if (cone_angle.defined() && cone_angle.numel()) { p.has_cone = 1; p.cone_angle = cone_angle.cpu().template item<float>(); }
		if (bounding_box.defined() && bounding_box.numel() == 6) { auto bb = host_floats(bounding_box); p.has_bbox = 1; for (int a = 0; a < 6; a++) p.bbox[a] = bb[a]; }
		p.seed = Seed; p.ray_base = RayCursor; RayCursor += n;
		nrf_render_outputs o{};
		o.d_rgb = res.Outputs.RGBMap.data_ptr<float>(); o.d_disp = res.Outputs.DispMap.data_ptr<float>(); o.d_acc = res.Outputs.AccMap.data_ptr<float>();
		o.d_depth = res.Outputs.DepthMap.data_ptr<float>();
		o.d_weights = return_weights ? res.Outputs.Weights.data_ptr<float>() : nullptr;
		o.d_raw = return_raw ? res.Raw.data_ptr<float>() : nullptr;
		const size_t wsb = nrf_render_rays_workspace_bytes(Renderer, n, &p);
		check(nrf_render_rays(Renderer, rays.data_ptr<float>(), stride, n, &p, t.data_ptr<float>(), u.defined() ? u.data_ptr<float>() : nullptr, &o,
			workspace(wsb, rays.device()), wsb, current_stream()), "nrf_render_rays");
		return res;
	}
};
#endif  // NRFPP_WITH_REFERENCE

}  // namespace nrfpp

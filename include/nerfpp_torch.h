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
//   TileComm                                     multi-GPU row tiles: RCCL all-gather of per-tile pixels behind the C ABI (no reference counterpart)
//   HipLeRFPass / HipLeRFRenderer : LeRFRenderer  the LeRF render pass (LeRFRenderer.h:56-132); the subclass needs -DNRFPP_WITH_LERF_RENDERER
//   HipNeRFRenderer<E, D, TNeRF> : NeRFRenderer<E, D, TNeRF>   overrides the virtuals Render / RenderRays / RunNetwork /
//                                                RawToOutputs (NeRFRenderer.h:96-158); Render is one library call per pose (nrf_render_rows);
//                                                BatchifyRays (the host chunk loop) stays inherited for callers that use it directly.
//                                                TNeRF is the reference's own NeRFSmall / NeRF module: its parameters are
//                                                read in named_parameters() order (SyncWeights() once; later changes are
//                                                picked up through ATen's version counters).  With grad mode on and
//                                                parameters that require grad, Render is ONE autograd node (RenderFn): the
//                                                reference's loop body -- Render, huber_loss, loss.backward(), Adam::step
//                                                (NeRFExecutor.h:862-995) -- trains through the HIP path unchanged.
//   HipHashEmbedderFunction                      the embedder's own autograd node (CuHashEmbedderFunction's counterpart)
//
// Compile inside the reference tree with -DNRFPP_WITH_REFERENCE (BaseEmbedder.h / NeRF.h / NeRFRenderer.h on the include
// path): the classes then derive from the reference's own bases and slot into NeRFExecutor<...>.
// Without it the header supplies a source-compatible BaseEmbedderImpl so that the encoders can be used on their own.
#pragma once

#include <torch/torch.h>
#include <c10/hip/HIPStream.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <functional>
#include <map>
#include <stdexcept>
#include <string>
#include <mutex>
#include <set>
#include <thread>
#include <vector>

#include "nerfpp_hip.h"
#include "nrf_rng.h"

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

/// Where a drop-in call spends its time (adapter_check bench): off by default -- one branch per scope.  When on, a scope drains the current stream on entry and on
/// exit, so that its wall time is its own host AND device work; the sum of the scopes is then an upper bound of the unsynchronised call.
struct PhaseClock {
	bool On = false;
	std::map<std::string, double> Ms;
	std::map<std::string, int64_t> Calls;
	void reset() { Ms.clear(); Calls.clear(); }
};
inline PhaseClock &phase_clock() { static PhaseClock c; return c; }
class PhaseScope {
	const char *Name;
	std::chrono::steady_clock::time_point T0;
	bool Live;
public:
	explicit PhaseScope(const char *name) : Name(name), Live(phase_clock().On) { if (Live) { c10::hip::getCurrentHIPStream().synchronize(); T0 = std::chrono::steady_clock::now(); } }
	~PhaseScope()
	{
		if (!Live) return;
		c10::hip::getCurrentHIPStream().synchronize();
		phase_clock().Ms[Name] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - T0).count();
		phase_clock().Calls[Name] += 1;
	}
};

/// The renderers that exist: an autograd node of HipNeRFRenderer::Render refers to its renderer by address and may outlive it (a graph kept past the renderer's
/// destruction); its backward asks here first and fails loudly instead of dereferencing a dangling pointer.
class LiveRenderers {
	std::mutex Mu;
	std::set<const void *> Set;
public:
	void add(const void *p) { std::lock_guard<std::mutex> lk(Mu); Set.insert(p); }
	void remove(const void *p) { std::lock_guard<std::mutex> lk(Mu); Set.erase(p); }
	bool alive(const void *p) { std::lock_guard<std::mutex> lk(Mu); return Set.count(p) != 0; }
};
inline LiveRenderers &live_renderers() { static LiveRenderers r; return r; }

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
	int NVolumes{1};
	/// NRF_HASH_CU (CuHashEmbedderImpl's state, CuHashEmbedder.h:14-27): the `<name>_embeddings` parameter [L * 2^T, F] and the four registered buffers.
	torch::Tensor Embeddings, Primes, Biases, FeatLocalSize, FeatLocalIdx;
	/// NRF_HASH_NGP (HashEmbedderImpl's state, NeRF.h:153): one nn::Embedding(2^T, F) per level, registered as `<name>_embeddings_<i>` (NeRF.cpp:255-259), so
	/// named_parameters() lists `<name>_embeddings_<i>.weight` and the reference's embedder_checkpoint.pt loads with torch::load.
	torch::nn::ModuleList LevelEmbeddings;

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
		const int64_t t_rows = (int64_t)1 << log2_hashmap_size;
		const auto cuda = torch::TensorOptions().device(torch::kCUDA);
		if (mode == NRF_HASH_CU) {
			// ---- CuHashEmbedderImpl's constructor state (CuHashEmbedder.cpp:24-76).  What must agree with the reference for torch::load interchange and for equal models
			// under equal seeds: the registered names, shapes and dtypes, and the ORDER in which the generators are drawn from -- the table first (one CUDA torch::rand),
			// then one CPU torch::randint per candidate multiplier until 3 * L * volumes primes in [2^28, 2^30) have been kept.
			Embeddings = register_parameter(module_name + "_embeddings", torch::rand({t_rows * NLevels, NFeaturesPerLevel}, cuda.dtype(torch::kFloat32)) * 1e-4f, /*requires_grad=*/true);
			Primes = DrawHashMultipliers(3 * NLevels * NVolumes).to(torch::kCUDA).reshape({NLevels, NVolumes, 3}).contiguous();
			Biases = torch::zeros({NLevels * NVolumes, 3}, cuda.dtype(torch::kFloat)).contiguous();          // RandBias is false there
			const int rows_per_level = (int)((((int64_t)1 << Log2HashmapSize) >> 4) << 4);                    // local_size, :63-68
			FeatLocalSize = torch::full({NLevels}, rows_per_level, cuda.dtype(torch::kInt32)).contiguous();
			FeatLocalIdx = (torch::arange(NLevels, cuda.dtype(torch::kInt32)) * rows_per_level).to(torch::kInt32).contiguous();
			Primes = register_buffer(module_name + "_primes", Primes);
			Biases = register_buffer(module_name + "_biases", Biases);
			FeatLocalSize = register_buffer(module_name + "_feat_local_size", FeatLocalSize);
			FeatLocalIdx = register_buffer(module_name + "_feat_local_idx", FeatLocalIdx);
		} else {
			// ---- HashEmbedderImpl (NeRF.cpp:255-271): L x nn::Embedding(2^T, F), registered per level, U(-1e-4, 1e-4) ----
			for (int i = 0; i < NLevels; i++) LevelEmbeddings->push_back(torch::nn::Embedding(t_rows, NFeaturesPerLevel));
			for (size_t i = 0; i < LevelEmbeddings->size(); i++) register_module(module_name + "_embeddings_" + std::to_string(i), LevelEmbeddings[i]);
			InitLevels();
			this->to(torch::kCUDA);
		}
	}
	~HipHashEmbedderImpl() override { nrf_hash_destroy(Handle); }

	/// `count` hash multipliers as an int32 CPU tensor: candidates are drawn one at a time from torch's CPU generator, uniform in [2^28, 2^30), and kept when prime
	/// (CuHashEmbedder.cpp:28-49 draws and tests them the same way, so equal seeds give equal multipliers).  Trial division by odd numbers up to the square root.
	static torch::Tensor DrawHashMultipliers(int count)
	{
		const auto cpu_i32 = torch::TensorOptions().dtype(torch::kInt32).device(torch::kCPU);
		auto out = torch::empty({count}, cpu_i32);
		int32_t *dst = out.data_ptr<int32_t>();
		for (int kept = 0; kept < count;) {
			const int64_t cand = torch::randint((int64_t)1 << 28, (int64_t)1 << 30, {1}, cpu_i32).item<int>();
			bool composite = cand % 2 == 0;
			for (int64_t q = 3; !composite && q * q <= cand; q += 2) composite = cand % q == 0;
			if (!composite) dst[kept++] = (int32_t)cand;
		}
		return out;
	}

	const nrf_hash *GetHandle() const { return Handle; }
	torch::Tensor GetBoundingBox() const { return BoundingBox; }
	int GetOutputDims() override { return NLevels * NFeaturesPerLevel; }
	int GetNLevels() const { return NLevels; }
	int GetNFeaturesPerLevel() const { return NFeaturesPerLevel; }
	int GetLog2HashmapSize() const { return Log2HashmapSize; }
	int GetBaseResolution() const { return BaseResolution; }
	int GetFinestResolution() const { return FinestResolution; }

	/// HashEmbedderImpl::Initialize (NeRF.cpp:264-271): custom uniform initialisation of every level
	void InitLevels() { for (size_t i = 0; i < LevelEmbeddings->size(); i++) for (auto p : LevelEmbeddings[i]->parameters()) torch::nn::init::uniform_(p, -0.0001, 0.0001); }

	/// The executor calls Initialize() on a freshly built embedder (NeRFExecutor.h:570; HashEmbedder re-draws its tables there, CuHashEmbedder's is empty) -- and the
	/// library needs the current table / primes pushed to it: both happen here.  Later changes of the parameters (an optimizer step, torch::load, copy_) are noticed by
	/// forward() itself through ATen's version counters (SyncIfChanged); after SetPrimes call Sync().
	void Initialize() { if (Mode == NRF_HASH_NGP) InitLevels(); Sync(); }
	void SetPrimes(torch::Tensor primes) { torch::NoGradGuard g; Primes.copy_(primes.view_as(Primes)); }
	/// the table in the layout nrf_hash_set_table takes: CU [L * 2^T, F]; NGP the levels' weights concatenated
	torch::Tensor Table()
	{
		if (Mode == NRF_HASH_CU) return Embeddings.detach();
		std::vector<torch::Tensor> lv;
		for (size_t i = 0; i < LevelEmbeddings->size(); i++) lv.push_back(LevelEmbeddings[i]->as<torch::nn::Embedding>()->weight.detach());
		return torch::cat(lv, 0);
	}
	/// torch::load re-homes parameters on the device they were SAVED from (a CPU-written checkpoint leaves them on the CPU; the executor follows its loads with
	/// ->to(device), NeRFExecutor.h:552-556): the upload below takes them from wherever they are.
	void Sync()
	{
		auto emb = dev_f32(Table().to(torch::kCUDA));
		check(nrf_hash_set_table(Handle, emb.data_ptr<float>(), 1, current_stream()), "nrf_hash_set_table");
		if (Mode == NRF_HASH_CU) {
			auto p = Primes.to(torch::kCPU, torch::kInt32).contiguous();
			auto b = host_floats(Biases);
			check(nrf_hash_set_primes(Handle, p.data_ptr<int32_t>(), b.data()), "nrf_hash_set_primes");
		}
		c10::hip::getCurrentHIPStream().synchronize();
		Synced = true; SyncedSignature = ParamSignature();
	}
	/// The table as ONE tensor in nrf_hash_set_table's layout, still attached to the parameters: CU the `<name>_embeddings` parameter itself, NGP the levels'
	/// weights concatenated (a differentiable torch::cat, so a gradient w.r.t. this tensor reaches every level's nn::Embedding weight).
	torch::Tensor TableForGrad()
	{
		if (Mode == NRF_HASH_CU) return Embeddings;
		std::vector<torch::Tensor> lv;
		for (size_t i = 0; i < LevelEmbeddings->size(); i++) lv.push_back(LevelEmbeddings[i]->as<torch::nn::Embedding>()->weight);
		return torch::cat(lv, 0);
	}
	bool AnyRequiresGrad() { for (auto &p : this->parameters()) if (p.requires_grad()) return true; return false; }
	/// changes whenever a parameter was written in place (optimizer step, copy_, torch::load) or replaced (->to(device)): ATen's version counters and the storage addresses
	uint64_t ParamSignature()
	{
		uint64_t sig = 1469598103934665603ull;
		for (auto &p : this->parameters()) { sig = (sig ^ (uint64_t)p._version()) * 1099511628211ull; sig = (sig ^ (uint64_t)(uintptr_t)p.data_ptr()) * 1099511628211ull; }
		return sig;
	}
	/// Upload the table when the parameters changed since the last upload -- device to device on the current stream, no host synchronisation.  Called by every
	/// forward, so a host that steps an optimizer over parameters() (NeRFExecutor::Train, NeRFExecutor.h:985) needs no extra call.
	void SyncIfChanged()
	{
		const uint64_t sig = ParamSignature();
		if (Synced && sig == SyncedSignature) return;
		PhaseScope ps("hash.sync_table");
		auto emb = dev_f32(Table().to(torch::kCUDA));
		check(nrf_hash_set_table(Handle, emb.data_ptr<float>(), 1, current_stream()), "nrf_hash_set_table");
		if (!Synced && Mode == NRF_HASH_CU) { auto p = Primes.to(torch::kCPU, torch::kInt32).contiguous(); auto b = host_floats(Biases); check(nrf_hash_set_primes(Handle, p.data_ptr<int32_t>(), b.data()), "nrf_hash_set_primes"); }
		Synced = true; SyncedSignature = sig;
	}
	/// Training keeps the table moving: of the baked dense pyramid of the render fast path (GBs, re-baked at every table upload) only the coarse levels that fit
	/// TrainDenseBudget stay baked while gradients flow (256 MB: ~0.1 ms of re-baking per step, and their lookups stay on the 2-load path: training step 9.3 -> 8.5 ms,
	/// same bits); the whole pyramid comes back with the first forward outside grad mode.
	int64_t TrainDenseBudget = (int64_t)256 << 20;
	void SetTraining(bool on)
	{
		if (on == TrainingMode) return;
		if (on) { DenseBudgetBefore = nrf_hash_get_dense_budget(Handle); check(nrf_hash_set_dense_budget(Handle, std::min<int64_t>(DenseBudgetBefore, TrainDenseBudget), current_stream()), "nrf_hash_set_dense_budget"); }
		else check(nrf_hash_set_dense_budget(Handle, DenseBudgetBefore, current_stream()), "nrf_hash_set_dense_budget");
		TrainingMode = on;
	}
	bool WantsGrad() { return torch::GradMode::is_enabled() && AnyRequiresGrad(); }

	std::pair<torch::Tensor, torch::Tensor> forward(torch::Tensor x) override;
private:
	bool Synced = false, TrainingMode = false;
	uint64_t SyncedSignature = 0;
	int64_t DenseBudgetBefore = 0;
};
TORCH_MODULE(HipHashEmbedder);

/// The autograd node of the embedder's forward: what CuHashEmbedderFunction is to CuHashEmbedderImpl::forward (CuHashEmbedder.cpp:85-103, CuHashEmbedder.cu:221-325).
/// forward: nrf_hash_encode on the table as uploaded; backward: nrf_hash_backward (CuHashEmbedderBackwardKernel's gradient / nn::Embedding's index_add, fp32) w.r.t. the
/// table tensor -- from where autograd carries it to `<name>_embeddings` / the per-level weights.  The query points get no gradient (as in the reference).
struct HipHashEmbedderFunction : public torch::autograd::Function<HipHashEmbedderFunction> {
	static torch::autograd::variable_list forward(torch::autograd::AutogradContext *ctx, torch::Tensor x, torch::Tensor table, int64_t handle)
	{
		auto *h = reinterpret_cast<nrf_hash *>(handle);
		x = dev_f32(x.detach()).view({-1, 3});
		auto out = torch::empty({x.size(0), (int64_t)nrf_hash_output_dims(h)}, x.options());
		auto mask = torch::empty({x.size(0)}, x.options().dtype(torch::kUInt8));
		check(nrf_hash_encode(h, x.data_ptr<float>(), x.size(0), out.data_ptr<float>(), mask.data_ptr<uint8_t>(), current_stream()), "nrf_hash_encode");
		ctx->save_for_backward({x});
		ctx->saved_data["handle"] = handle;
		ctx->saved_data["table_sizes"] = table.sizes().vec();
		auto keep = mask.to(torch::kBool);
		ctx->mark_non_differentiable({keep});
		return {out, keep};
	}
	static torch::autograd::variable_list backward(torch::autograd::AutogradContext *ctx, torch::autograd::variable_list grads)
	{
		auto *h = reinterpret_cast<nrf_hash *>(ctx->saved_data["handle"].toInt());
		auto x = ctx->get_saved_variables()[0];
		auto g_table = torch::zeros(ctx->saved_data["table_sizes"].toIntVector(), x.options());
		if (grads[0].defined()) {
			auto g = dev_f32(grads[0]);
			check(nrf_hash_backward(h, x.data_ptr<float>(), x.size(0), g.data_ptr<float>(), g_table.data_ptr<float>(), current_stream()), "nrf_hash_backward");
		}
		return {torch::Tensor(), g_table, torch::Tensor()};
	}
};

inline std::pair<torch::Tensor, torch::Tensor> HipHashEmbedderImpl::forward(torch::Tensor x)
{
	const bool grad = WantsGrad();
	SetTraining(grad);
	SyncIfChanged();
	if (grad) {
		auto r = HipHashEmbedderFunction::apply(x, TableForGrad(), (int64_t)reinterpret_cast<intptr_t>(Handle));
		return std::make_pair(r[0], r[1]);
	}
	x = dev_f32(x).view({-1, 3});
	auto out = torch::empty({x.size(0), GetOutputDims()}, x.options());
	auto mask = torch::empty({x.size(0)}, x.options().dtype(torch::kUInt8));
	check(nrf_hash_encode(Handle, x.data_ptr<float>(), x.size(0), out.data_ptr<float>(), mask.data_ptr<uint8_t>(), current_stream()), "nrf_hash_encode");
	return std::make_pair(out, mask.to(torch::kBool));
}

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
	/// `launch_tag` must be the SAME string on every rank of one launch and DIFFERENT between launches that share `id_path` (a job id, the launcher's start
	/// time, MASTER_PORT): the file carries it in its header and a reader ignores a file with another tag -- so a file left behind by an earlier run is never
	/// taken for this run's id (with an empty tag, `id_path` itself must be unique per launch; rank 0 then removes any existing file first).  Rank 0 unlinks the
	/// file once the communicator exists (every rank has read it by then).  Both the wait for the file and the communicator's own rendezvous are bounded by
	/// `timeout_s` (nrf_comm_create_timeout): a peer that never arrives raises instead of parking this rank for ever.
	TileComm(int world, int rank, const std::string &id_path, double timeout_s = 120.0, const std::string &launch_tag = "") : World(world), Rank(rank)
	{
		unsigned char id[NRF_COMM_ID_BYTES];
		char tag[64] = {0};
		std::snprintf(tag, sizeof(tag), "%s", launch_tag.c_str());
		if (rank == 0) {
			check(nrf_comm_unique_id(id), "nrf_comm_unique_id");
			if (world > 1) {
				std::remove(id_path.c_str());                            // a stale file of an earlier launch
				const std::string tmp = id_path + ".tmp";
				{ std::ofstream f(tmp, std::ios::binary); f.write(tag, sizeof(tag)); f.write(reinterpret_cast<const char *>(id), sizeof(id)); }
				if (std::rename(tmp.c_str(), id_path.c_str()) != 0) throw std::runtime_error("TileComm: cannot publish the communicator id at " + id_path);
			}
		} else {
			const auto t0 = std::chrono::steady_clock::now();
			for (;;) {
				std::ifstream f(id_path, std::ios::binary);
				char got[64];
				if (f && f.read(got, sizeof(got)) && f.read(reinterpret_cast<char *>(id), sizeof(id)) && f.gcount() == (std::streamsize)sizeof(id) &&
					std::memcmp(got, tag, sizeof(tag)) == 0) break;
				if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s)
					throw std::runtime_error("TileComm: no communicator id of this launch at " + id_path);
				std::this_thread::sleep_for(std::chrono::milliseconds(20));
			}
		}
		check(nrf_comm_create_timeout(id, world, rank, timeout_s, &Comm), "nrf_comm_create");
		if (rank == 0 && world > 1) std::remove(id_path.c_str());
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
	/// Data-parallel training (no reference counterpart): between `loss.backward()` and `Optimizer->step()` of NeRFExecutor::Train's loop body
	/// (NeRFExecutor.h:923 / :985) every rank hands the gradients of its parameters over -- they become their mean over the ranks IN PLACE (nrf_allreduce_grads: bucketed
	/// ncclAllReduce + one scale on the current stream), so all replicas take the same Adam step.  Parameters without a gradient are given a zero one first (all ranks must
	/// pass the same buffers).  overflow: this rank's fp16-backward overflow report (HipNeRFRenderer redoes such a step in fp32 by itself: pass false); the return value is
	/// the ranks' agreement -- true on every rank iff any rank passed true, and nothing was exchanged then (skip the optimizer step).
	bool AllReduceGrads(std::vector<torch::Tensor> params, bool overflow = false, int64_t bucket_bytes = (int64_t)32 << 20) const
	{
		std::vector<float *> ptrs; std::vector<int64_t> counts; std::vector<torch::Tensor> keep;
		for (auto &p : params) {
			if (!p.defined() || p.numel() == 0) continue;
			if (!p.grad().defined()) p.mutable_grad() = torch::zeros_like(p);
			TORCH_CHECK(p.grad().is_cuda() && p.grad().scalar_type() == torch::kFloat32, "AllReduceGrads: gradients must be fp32 tensors on the GPU");
			if (!p.grad().is_contiguous()) p.mutable_grad() = p.grad().contiguous();
			keep.push_back(p.grad()); ptrs.push_back(keep.back().data_ptr<float>()); counts.push_back(keep.back().numel());
		}
		int skip = 0;
		check(nrf_allreduce_grads(Comm, ptrs.data(), counts.data(), (int)ptrs.size(), bucket_bytes, overflow ? 1 : 0, &skip, current_stream()), "nrf_allreduce_grads");
		return skip != 0;
	}
};

// ---------------------------------------------------------------------------------------------------------------------
// HipAdam : torch::optim::Adam -- the executor's optimizer (NeRFExecutor.h:539: `torch::optim::Adam(grad_vars, AdamOptions(lr).eps(1e-15).betas({0.9, 0.99}))`) with
// its step as ONE kernel per parameter (nrf_adam_step: 4 reads + 3 writes per element) instead of LibTorch's chain of elementwise passes over the 64 MiB table.
// Same update rule, same per-parameter state objects (torch::optim::AdamParamState: step, exp_avg, exp_avg_sq), so torch::save / torch::load of the optimizer
// (optimizer_checkpoint.pt, NeRFExecutor.h:1055-1070) interchange with the reference's.  Options the kernel does not implement (weight_decay, amsgrad), parameters off the
// GPU or not fp32 take torch::optim::Adam::step for the whole call.
// ---------------------------------------------------------------------------------------------------------------------
class HipAdam : public torch::optim::Adam {
public:
	using torch::optim::Adam::Adam;
	torch::Tensor step(LossClosure closure = nullptr) override
	{
		for (auto &group : param_groups_) {
			auto &o = static_cast<torch::optim::AdamOptions &>(group.options());
			bool plain = o.weight_decay() == 0 && !o.amsgrad();
			for (auto &p : group.params()) if (p.grad().defined() && !(p.is_cuda() && p.scalar_type() == torch::kFloat32 && p.is_contiguous() && !p.grad().is_sparse())) plain = false;
			if (!plain) return torch::optim::Adam::step(closure);
		}
		torch::NoGradGuard ng;
		torch::Tensor loss;
		if (closure != nullptr) { at::AutoGradMode enable_grad(true); loss = closure(); }
		for (auto &group : param_groups_) {
			auto &o = static_cast<torch::optim::AdamOptions &>(group.options());
			for (auto &p : group.params()) {
				if (!p.grad().defined()) continue;
				auto it = state_.find(p.unsafeGetTensorImpl());
				if (it == state_.end()) {
					auto st = std::make_unique<torch::optim::AdamParamState>();
					st->step(0);
					st->exp_avg(torch::zeros_like(p, torch::MemoryFormat::Preserve));
					st->exp_avg_sq(torch::zeros_like(p, torch::MemoryFormat::Preserve));
					it = state_.emplace(p.unsafeGetTensorImpl(), std::move(st)).first;
				}
				auto &st = static_cast<torch::optim::AdamParamState &>(*it->second);
				st.step(st.step() + 1);
				auto g = p.grad().to(torch::kFloat32).contiguous();
				check(nrf_adam_step(p.data_ptr<float>(), g.data_ptr<float>(), st.exp_avg().data_ptr<float>(), st.exp_avg_sq().data_ptr<float>(), p.numel(), (float)o.lr(),
					(float)std::get<0>(o.betas()), (float)std::get<1>(o.betas()), (float)o.eps(), (int)st.step(), current_stream()), "nrf_adam_step");
				p.unsafeGetTensorImpl()->bump_version();          // written behind ATen's back: the drop-in's SyncIfChanged watches the version counters
			}
		}
		return loss;
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

// ---------------------------------------------------------------------------------------------------------------------
// LeRF render pass (LeRFRenderer.h:56-132, LeRFRenderer.cpp): CuHashEmbedder features -> LeRF head -> sigma_le weights -> rendered CLIP embedding.
// HipLeRFPass holds everything that does not need the reference's headers (so it links and runs without LeRFRenderer.cpp, which pulls in the external
// RuCLIP module for `Relevancy`); HipLeRFRenderer below subclasses the reference's LeRFRenderer onto it.
// ---------------------------------------------------------------------------------------------------------------------
struct LeRFPassOutputs {                 // LeRFRendererOutputs (LeRFRenderer.h:9-18) -- same member names.  Relevancy: [N, 2], defined when prompts are set
	torch::Tensor LangEmbedding, RenderedLangEmbedding, DispMapLE, AccMapLE, WeightsLE, DepthMapLE, Relevancy;
};

/// Relevancy(embeds [N, E], positives [P, E], negatives [Q, E]) -> [N, 2] (call sites LeRFRenderer.cpp:79, NeRFExecutor.h:824).  The reference takes this function from
/// the external RuCLIP module (RuCLIPProcessor.h); this is the published LERF relevancy score it mirrors -- PARITY UNPINNED (nerfpp_hip.h, nrf_lerf_relevancy).
inline torch::Tensor Relevancy(torch::Tensor embeds, torch::Tensor positives, torch::Tensor negatives, int positive_id = 0)
{
	auto e = dev_f32(embeds);
	auto pos = positives.to(e.device(), torch::kFloat32).reshape({-1, e.size(1)}).contiguous(), neg = negatives.to(e.device(), torch::kFloat32).reshape({-1, e.size(1)}).contiguous();
	auto out = torch::empty({e.size(0), 2}, e.options());
	check(nrf_lerf_relevancy(e.data_ptr<float>(), e.size(0), (int)e.size(1), pos.data_ptr<float>(), (int)pos.size(0), neg.data_ptr<float>(), (int)neg.size(0), positive_id,
		out.data_ptr<float>(), current_stream()), "nrf_lerf_relevancy");
	return out;
}

/// The relevancy picture of RenderPath (NeRFExecutor.h:713-719): rel[..., 0].mul(255).to(kU8) -> COLORMAP_JET, [..., 3] bytes in OpenCV's B, G, R order (parity unpinned)
inline torch::Tensor RelevancyImage(torch::Tensor relevancy)
{
	auto r = dev_f32(relevancy);
	auto sz = r.sizes().vec(); const int stride = (int)sz.back(); sz.back() = 3;
	const int64_t n = r.numel() / stride;
	auto out = torch::empty(sz, r.options().dtype(torch::kUInt8));
	check(nrf_relevancy_image(r.data_ptr<float>(), n, stride, out.data_ptr<uint8_t>(), current_stream()), "nrf_relevancy_image");
	return out;
}

/// nrf_mlp_small_desc of a LeRF module (LeRF.cpp:3-26) from its parameter shapes, named_parameters() order: sigma_le_net_0..L-1, le_net_0..L-1
template <class TLeRF>
inline nrf_mlp_small_desc lerf_desc_of(TLeRF &lerf)
{
	std::vector<torch::Tensor> w;
	for (auto &p : lerf->named_parameters()) w.push_back(p.value());
	TORCH_CHECK(w.size() >= 2 && w.size() % 2 == 0, "LeRF: expected 2 x num_layers_le bias-free Linear weights");
	const int nl = (int)w.size() / 2;
	nrf_mlp_small_desc d{};
	d.input_ch = (int)w[0].size(1); d.input_ch_views = 0; d.num_layers = nl; d.hidden_dim = (int)w[0].size(0);
	d.geo_feat_dim = (int)w[nl - 1].size(0) - 1; d.num_layers_color = nl; d.hidden_dim_color = (int)w[2 * nl - 1].size(0);
	return d;
}

class HipLeRFPass {
	HipHashEmbedder LangEmbedFn = nullptr;
	MlpHandle Mlp;
	nrf_mlp_small_desc Desc{};
	int Precision;
	bool Fused = false, LevelMajor = false;
	nrf_lerf_renderer *Pass = nullptr;       // the render pass as library calls (nrf_lerf_render_rays / _batchify_rays / _render_rows), level-major fused configuration
	torch::Tensor Workspace, Positives, Negatives;
	std::vector<std::pair<int, torch::Tensor>> LinCache;
	torch::Tensor linspace01(int steps, torch::Device dev)
	{
		for (auto &e : LinCache) if (e.first == steps && e.second.device() == dev) return e.second;
		LinCache.emplace_back(steps, torch::linspace(0.f, 1.f, steps, torch::kFloat).to(dev));
		return LinCache.back().second;
	}
	void *workspace(size_t bytes, torch::Device dev)
	{
		if (!Workspace.defined() || (size_t)Workspace.numel() < bytes || Workspace.device() != dev) Workspace = torch::empty({(int64_t)bytes}, torch::TensorOptions().dtype(torch::kUInt8).device(dev));
		return Workspace.data_ptr();
	}
	bool SingleCallOk(int s, int ni) const { return SingleCall && Pass && Fused && LevelMajor && ReuseFeatures && HandOverGeo && ni > 0 && s % 32 == 0 && (s + ni) % 32 == 0; }
	/// outputs of a single-call render over n rays; `o` receives the tensors, `ro` their addresses
	void alloc_outputs(int64_t n, int sf, bool return_weights, torch::TensorOptions opt, LeRFPassOutputs &o, nrf_lerf_outputs &ro, torch::Tensor *z_fine)
	{
		o.DispMapLE = torch::empty({n}, opt); o.AccMapLE = torch::empty({n}, opt); o.DepthMapLE = torch::empty({n}, opt);
		ro.d_disp = o.DispMapLE.data_ptr<float>(); ro.d_acc = o.AccMapLE.data_ptr<float>(); ro.d_depth = o.DepthMapLE.data_ptr<float>();
		if (return_weights) {                                   // LeRFRenderer.cpp:180-185: without ReturnWeights the weights and the rendered embedding are dropped
			o.WeightsLE = torch::empty({n, (int64_t)sf}, opt); o.RenderedLangEmbedding = torch::empty({n, (int64_t)GetLangEmbedDim()}, opt);
			ro.d_weights = o.WeightsLE.data_ptr<float>(); ro.d_embedding = o.RenderedLangEmbedding.data_ptr<float>();
		}
		if (Positives.defined() && Negatives.defined()) { o.Relevancy = torch::empty({n, 2}, opt); ro.d_relevancy = o.Relevancy.data_ptr<float>(); }      // LeRFRenderer.cpp:79
		if (z_fine) { *z_fine = torch::empty({n, (int64_t)sf}, opt); ro.d_z_fine = z_fine->data_ptr<float>(); }
	}
	nrf_render_params pass_params(int s, int ni, bool lin_disp) const
	{
		nrf_render_params p{};
		p.n_samples = s; p.n_importance = ni; p.lindisp = lin_disp; p.precision = Precision; p.sum_vec = 8;
		p.coarse_mode = ExactCoarse ? NRF_COARSE_AUTO : NRF_COARSE_FULL;
		return p;
	}
public:
	HipLeRFPass(const HipLeRFPass &) = delete;
	HipLeRFPass &operator=(const HipLeRFPass &) = delete;
	~HipLeRFPass() { live_renderers().remove(this); nrf_lerf_renderer_destroy(Pass); }
	bool SingleCall = true;             ///< the pass as library calls (false: stage-composed below, for A/B tests)
	bool ReuseRenderFeatures = true;    ///< the training backward reads the language features its forward render left in the workspace (nrf_lerf_renderer_last_features) when it can
	bool ReusedRenderFeatures = false;  ///< ... and whether the last backward did

	/// LeRFRenderer::SetLeRFPrompts (LeRFRenderer.h:86): [P, E] positive and [Q, E] negative phrase embeddings from the host's text encoder; undefined tensors clear them.
	/// With prompts set every render fills Relevancy.
	void SetLeRFPrompts(torch::Tensor positives, torch::Tensor negatives)
	{
		Positives = positives; Negatives = negatives;
		PushPrompts();
	}
	void PushPrompts()
	{
		if (!Pass) return;
		if (!(Positives.defined() && Negatives.defined() && Positives.numel() && Negatives.numel())) { Positives = torch::Tensor(); Negatives = torch::Tensor(); check(nrf_lerf_set_prompts(Pass, nullptr, 0, nullptr, 0, 0, current_stream()), "nrf_lerf_set_prompts"); return; }
		auto pos = host_floats(Positives), neg = host_floats(Negatives);
		const int E = GetLangEmbedDim();
		check(nrf_lerf_set_prompts(Pass, pos.data(), (int)pos.size() / E, neg.data(), (int)neg.size() / E, 0, current_stream()), "nrf_lerf_set_prompts");
	}

	/// precision of the fused matrix-core passes: NRF_PREC_F16_SPLIT (fp32-grade, as LeRFImpl::forward computes) or NRF_PREC_F16_MFMA
	explicit HipLeRFPass(HipHashEmbedder lang_embed_fn, int precision = NRF_PREC_F16_SPLIT) : LangEmbedFn(lang_embed_fn), Precision(precision) { live_renderers().add(this); }

	int GetLangEmbedDim() const { return Desc.hidden_dim_color; }
	bool IsFused() const { return Fused; }

	/// (Re)read the LeRF head's parameters; call after construction, checkpoint load or an optimizer step.
	void SyncWeights(const nrf_mlp_small_desc &d, const std::vector<float> &blob)
	{
		TORCH_CHECK((int64_t)blob.size() == nrf_mlp_lerf_param_count(&d), "LeRF parameter count mismatch");
		nrf_mlp *m = nullptr;
		check(nrf_mlp_lerf_create(&d, blob.data(), 0, current_stream(), &m), "nrf_mlp_lerf_create");
		Mlp.reset(m); Desc = d;
		LangEmbedFn->Sync();
		Fused = nrf_lerf_mfma_available(m) != 0;
		if (Fused) check(nrf_lerf_set_precision(m, Precision), "nrf_lerf_set_precision");
		LevelMajor = Fused && LangEmbedFn->Mode == NRF_HASH_CU && LangEmbedFn->NLevels == 16 && LangEmbedFn->NFeaturesPerLevel == 8;
		nrf_lerf_renderer_destroy(Pass); Pass = nullptr;
		if (LevelMajor) {
			nrf_lerf_renderer_desc rd{LangEmbedFn->GetHandle(), Mlp.m};
			check(nrf_lerf_renderer_create(&rd, &Pass), "nrf_lerf_renderer_create");
			PushPrompts();
		}
	}
	template <class TLeRF> void SyncWeights(TLeRF &lerf)
	{
		SyncWeights(lerf_desc_of(lerf), parameter_blob(lerf));
		// the module's parameters as one blob in named_parameters() order (== the head's blob layout), still attached to them: a gradient w.r.t. this tensor reaches every
		// Linear weight through torch::cat's backward; and their version counters / storage addresses, so that an optimizer step is noticed without a call from the host
		BlobForGradFn = [lerf]() mutable { std::vector<torch::Tensor> flat; for (auto &p : lerf->named_parameters()) flat.push_back(p.value().reshape({-1})); return torch::cat(flat, 0); };
		HeadSignatureFn = [lerf]() mutable {
			uint64_t sig = 1469598103934665603ull;
			for (auto &p : lerf->parameters()) { sig = (sig ^ (uint64_t)p._version()) * 1099511628211ull; sig = (sig ^ (uint64_t)(uintptr_t)p.data_ptr()) * 1099511628211ull; }
			return sig;
		};
		HeadRequiresGradFn = [lerf]() mutable { for (auto &p : lerf->parameters()) if (p.requires_grad()) return true; return false; };
		HeadSignature = HeadSignatureFn();
	}

	// ---- the training render: LeRFRenderer::Render on a RAY BATCH as ONE autograd node (NeRFExecutor.h:955-982: Render -> huber(...).sum(-1).nanmean() -> lang_loss.backward()) ----
	std::function<torch::Tensor()> BlobForGradFn;
	std::function<uint64_t()> HeadSignatureFn;
	std::function<bool()> HeadRequiresGradFn;
	uint64_t HeadSignature = 0;
	torch::Tensor LastFineDepths, LastRays;      ///< of the most recent training render: z_vals of the fine pass [n, S + N_importance] and the packed rays [n, 8 | 11]
	bool WantsGrad() { return torch::GradMode::is_enabled() && ((HeadRequiresGradFn && HeadRequiresGradFn()) || LangEmbedFn->AnyRequiresGrad()); }
	/// Upload the head's parameters / the language table when they changed since the last upload (an optimizer step over the module's own parameters, torch::load, copy_)
	void SyncIfChanged()
	{
		if (HeadSignatureFn) {
			const uint64_t sig = HeadSignatureFn();
			if (sig != HeadSignature) {
				torch::NoGradGuard ng;
				auto blob = dev_f32(BlobForGradFn().to(torch::kCUDA));
				check(nrf_mlp_set_params(Mlp.m, blob.data_ptr<float>(), 1, current_stream()), "nrf_mlp_set_params");
				if (Fused) check(nrf_lerf_set_precision(Mlp.m, Precision), "nrf_lerf_set_precision");
				HeadSignature = sig;
			}
		}
		LangEmbedFn->SyncIfChanged();
	}

	/// forward: the library's Chunk loop over the packed rays (nrf_lerf_batchify_rays: the fused matrix-core passes, raw_le never formed) keeping the fine depths;
	/// backward: ONE library call (nrf_lerf_backward_points: language-grid encode, the head's recomputed fp32 forward and backward, the grid's scatter) w.r.t. the language
	/// table and the head's parameter blob.  Gradients flow from RenderedLangEmbedding only (what lang_loss reads, NeRFExecutor.h:970-974); z_samples are detached
	/// (LeRFRenderer.cpp:150), so the coarse pass carries none.
	struct LeRFRenderFn : public torch::autograd::Function<LeRFRenderFn> {
		static torch::autograd::variable_list forward(torch::autograd::AutogradContext *ctx, torch::Tensor rays_, torch::Tensor table, torch::Tensor blob, int64_t self_i, int64_t s,
			int64_t ni, int64_t chunk, bool lin_disp)
		{
			auto *self = reinterpret_cast<HipLeRFPass *>(self_i);
			const int64_t n = rays_.size(0); const int stride = (int)rays_.size(1);
			const auto opt = rays_.options();
			LeRFPassOutputs out; nrf_lerf_outputs ro{}; torch::Tensor zf;
			self->alloc_outputs(n, (int)(s + ni), true, opt, out, ro, &zf);
			nrf_render_params p = self->pass_params((int)s, (int)ni, lin_disp);
			if (n > 0) {
				const size_t wsb = nrf_lerf_batchify_rays_workspace_bytes(self->Pass, n, (int)chunk, &p);
				check(nrf_lerf_batchify_rays(self->Pass, rays_.data_ptr<float>(), stride, n, (int)chunk, &p, self->linspace01((int)s, rays_.device()).template data_ptr<float>(),
					self->linspace01((int)ni, rays_.device()).template data_ptr<float>(), &ro, self->workspace(wsb, rays_.device()), wsb, current_stream()), "nrf_lerf_batchify_rays");
			}
			ctx->save_for_backward({rays_, zf});
			ctx->saved_data["self"] = self_i;
			{	// where this render left the language features of its fine depths (a one-chunk call): the backward reads them if no render has touched the workspace since
				const void *f = nullptr; const uint8_t *k = nullptr; const int32_t *sr = nullptr; int64_t cols = 0, vn = 0; int vsf = 0; uint64_t serial = 0;
				const int rcv = n > 0 ? nrf_lerf_renderer_last_features(self->Pass, &f, &cols, &k, &sr, &vn, &vsf, &serial) : NRF_ERR_UNSUPPORTED;
				ctx->saved_data["view_serial"] = (rcv == NRF_OK && vn == n && vsf == (int)(s + ni)) ? (int64_t)serial : (int64_t)-1;
			}
			ctx->saved_data["table_sizes"] = table.sizes().vec();
			ctx->saved_data["blob_numel"] = blob.numel();
			std::vector<torch::Tensor> nd{out.WeightsLE, out.DepthMapLE, out.DispMapLE, out.AccMapLE, zf};
			if (out.Relevancy.defined()) nd.push_back(out.Relevancy);
			ctx->mark_non_differentiable(nd);
			return {out.RenderedLangEmbedding, out.WeightsLE, out.DepthMapLE, out.DispMapLE, out.AccMapLE, out.Relevancy.defined() ? out.Relevancy : torch::empty({0}, opt), zf};
		}
		static torch::autograd::variable_list backward(torch::autograd::AutogradContext *ctx, torch::autograd::variable_list grads)
		{
			auto &sd = ctx->saved_data;
			const int64_t self_i = sd["self"].toInt();
			TORCH_CHECK(self_i != 0 && live_renderers().alive(reinterpret_cast<const void *>(self_i)), "HipLeRFPass: backward through a Render() whose pass has been destroyed");
			auto *self = reinterpret_cast<HipLeRFPass *>(self_i);
			auto saved = ctx->get_saved_variables();
			auto rays = saved[0], z = saved[1];
			const int64_t n = rays.size(0); const int stride = (int)rays.size(1), s = (int)z.size(1);
			const auto opt = rays.options();
			auto g_table = torch::zeros(sd["table_sizes"].toIntVector(), opt), g_blob = torch::zeros({sd["blob_numel"].toInt()}, opt);
			if (n > 0 && grads[0].defined()) {
				auto g = dev_f32(grads[0]).reshape({n, -1});
				auto pts = torch::empty({n * s, 3}, opt);
				check(nrf_points(rays.data_ptr<float>(), stride, z.data_ptr<float>(), n, s, pts.data_ptr<float>(), current_stream()), "nrf_points");
				const size_t wsb = nrf_lerf_backward_points_workspace_bytes(self->Pass, n, s);
				auto ws = torch::empty({(int64_t)wsb}, opt.dtype(torch::kUInt8));          // not the pass's render workspace: a backward may run while another render is being issued
				const void *vf = nullptr; const uint8_t *vk = nullptr; const int32_t *vs = nullptr; int64_t vcols = 0, vn = 0; int vsf = 0; uint64_t serial = 0;
				const int64_t want = sd["view_serial"].toInt();
				const bool view = self->ReuseRenderFeatures && want >= 0 && nrf_lerf_renderer_last_features(self->Pass, &vf, &vcols, &vk, &vs, &vn, &vsf, &serial) == NRF_OK &&
					(int64_t)serial == want && vn == n && vsf == s;
				self->ReusedRenderFeatures = view;
				if (view)
					check(nrf_lerf_backward_points_src(self->Pass, vf, vcols, vk, vs, pts.data_ptr<float>(), z.data_ptr<float>(), rays.data_ptr<float>() + 3, stride, n, s, nullptr, 0.f,
						g.data_ptr<float>(), g_blob.data_ptr<float>(), g_table.data_ptr<float>(), ws.data_ptr(), wsb, current_stream()), "nrf_lerf_backward_points_src");
				else
				check(nrf_lerf_backward_points(self->Pass, pts.data_ptr<float>(), z.data_ptr<float>(), rays.data_ptr<float>() + 3, stride, n, s, nullptr, 0.f, g.data_ptr<float>(),
					g_blob.data_ptr<float>(), g_table.data_ptr<float>(), ws.data_ptr(), wsb, current_stream()), "nrf_lerf_backward_points");
			}
			return {torch::Tensor(), g_table, g_blob, torch::Tensor(), torch::Tensor(), torch::Tensor(), torch::Tensor(), torch::Tensor()};
		}
	};

	/// LeRFRenderer::Render (LeRFRenderer.cpp:265-330) on an explicit ray batch, deterministic settings (ThinRay, Perturb = RawNoiseStd = StochasticPreconditioningAlpha = 0):
	/// AABB clipping and packing, then the Chunk loop.  With grad mode on and parameters that require grad the result is differentiable (LeRFRenderFn); the head's and the
	/// table's current values are picked up by themselves (SyncIfChanged).
	LeRFPassOutputs RenderBatch(torch::Tensor rays_o, torch::Tensor rays_d, torch::Tensor bounding_box, const int n_samples, const int n_importance, const int chunk,
		const bool lin_disp = false, const bool return_weights = true, float *near_out = nullptr, float *far_out = nullptr)
	{
		auto o = dev_f32(rays_o).reshape({-1, 3}).contiguous(), d = dev_f32(rays_d).reshape({-1, 3}).contiguous();
		const int64_t n = o.size(0);
		const int stride = 8;                                                   // LeRFRenderer::Render packs [o, d, near, far] only (LeRFRenderer.cpp:298)
		auto bb = host_floats(bounding_box);
		TORCH_CHECK(bb.size() == 6, "RenderBatch: bounding_box must hold [min xyz, max xyz]");
		auto rays_ = torch::empty({n, stride}, o.options());
		if (n > 0) check(nrf_pack_rays(o.data_ptr<float>(), d.data_ptr<float>(), bb.data(), n, 0, rays_.data_ptr<float>(), current_stream()), "nrf_pack_rays");
		const bool train = WantsGrad();
		LangEmbedFn->SetTraining(train);
		SyncIfChanged();
		LeRFPassOutputs out;
		if (train) {
			TORCH_CHECK(SingleCallOk(n_samples, n_importance) && BlobForGradFn, "training through HipLeRFPass needs the library path (CuHashEmbedder L16 F8 language grid, a head of the "
				"built family, hierarchical sampling with sample counts in multiples of 32) and SyncWeights(lerf module)");
			auto r = LeRFRenderFn::apply(rays_, LangEmbedFn->TableForGrad(), BlobForGradFn(), (int64_t)reinterpret_cast<intptr_t>(this), (int64_t)n_samples, (int64_t)n_importance,
				(int64_t)chunk, lin_disp);
			out.RenderedLangEmbedding = r[0]; out.WeightsLE = r[1]; out.DepthMapLE = r[2]; out.DispMapLE = r[3]; out.AccMapLE = r[4];
			if (r[5].numel()) out.Relevancy = r[5];
			LastFineDepths = r[6]; LastRays = rays_;
			if (!return_weights) { out.WeightsLE = torch::Tensor(); out.RenderedLangEmbedding = torch::Tensor(); }      // LeRFRenderer.cpp:180-185
		} else {
			std::vector<torch::Tensor> e, w_, dep, disp, acc, rel;
			for (int64_t i = 0; i < n; i += chunk) {
				auto part = RenderRays(rays_.index({torch::indexing::Slice(i, std::min<int64_t>(i + chunk, n))}), n_samples, lin_disp, n_importance, return_weights);
				if (part.RenderedLangEmbedding.defined()) e.push_back(part.RenderedLangEmbedding);
				if (part.WeightsLE.defined()) w_.push_back(part.WeightsLE);
				if (part.Relevancy.defined()) rel.push_back(part.Relevancy);
				dep.push_back(part.DepthMapLE); disp.push_back(part.DispMapLE); acc.push_back(part.AccMapLE);
			}
			if (!e.empty()) out.RenderedLangEmbedding = torch::cat(e, 0);
			if (!w_.empty()) out.WeightsLE = torch::cat(w_, 0);
			if (!rel.empty()) out.Relevancy = torch::cat(rel, 0);
			if (!dep.empty()) { out.DepthMapLE = torch::cat(dep, 0); out.DispMapLE = torch::cat(disp, 0); out.AccMapLE = torch::cat(acc, 0); }
		}
		if ((near_out || far_out) && n > 0) {
			float nr = 0.f, fr = 0.f;
			check(nrf_near_far_range(rays_.data_ptr<float>(), n, stride, &nr, &fr, current_stream()), "nrf_near_far_range");          // LeRFRenderer.cpp:327-328
			if (near_out) *near_out = nr;
			if (far_out) *far_out = fr;
		}
		return out;
	}

	/// LeRFRenderer::RunLENetwork (LeRFRenderer.cpp:5-25): [N,S,3] -> [N,S,E+1] in fp32, sigma_le zeroed where the embedder's keep_mask is false
	torch::Tensor RunLENetwork(torch::Tensor inputs)
	{
		auto pts = dev_f32(inputs);
		auto flat = pts.view({-1, 3});
		auto [emb, keep] = LangEmbedFn->forward(flat);
		auto out = torch::empty({flat.size(0), (int64_t)nrf_mlp_output_dims(Mlp.m)}, emb.options());
		check(nrf_mlp_forward(Mlp.m, emb.data_ptr<float>(), flat.size(0), NRF_PREC_F32, out.data_ptr<float>(), current_stream()), "nrf_mlp_forward");
		out.index_put_({~keep, -1}, 0);
		auto sz = pts.sizes().vec(); sz.back() = out.size(1);
		return out.view(sz);
	}

	/// LeRFRenderer::RawToLEOutputs (LeRFRenderer.cpp:27-82) without Relevancy
	LeRFPassOutputs RawToLEOutputs(torch::Tensor raw_le, torch::Tensor z_vals_le, torch::Tensor rays_d, const int lang_embed_dim = 768, const float raw_noise_std = 0.f)
	{
		TORCH_CHECK(raw_noise_std == 0.f, "RawToLEOutputs: raw_noise_std > 0 is the training-time noise branch (torch::randn_like); not built");
		auto raw = dev_f32(raw_le); auto z = dev_f32(z_vals_le); auto d = dev_f32(rays_d);
		const int64_t n = raw.size(0); const int s = (int)raw.size(1), c = (int)raw.size(2);
		LeRFPassOutputs o;
		using torch::indexing::Slice;
		o.LangEmbedding = raw.index({"...", Slice(0, lang_embed_dim)});
		o.WeightsLE = torch::empty({n, s}, raw.options()); o.DepthMapLE = torch::empty({n}, raw.options()); o.DispMapLE = torch::empty({n}, raw.options()); o.AccMapLE = torch::empty({n}, raw.options());
		check(nrf_raw2weights(raw.data_ptr<float>(), c, lang_embed_dim, z.data_ptr<float>(), d.data_ptr<float>(), 3, n, s, o.WeightsLE.data_ptr<float>(), o.DepthMapLE.data_ptr<float>(),
			o.DispMapLE.data_ptr<float>(), o.AccMapLE.data_ptr<float>(), current_stream()), "nrf_raw2weights");
		o.RenderedLangEmbedding = torch::empty({n, (int64_t)lang_embed_dim}, raw.options());
		check(nrf_render_clip_embedding(raw.data_ptr<float>(), c, lang_embed_dim, o.WeightsLE.data_ptr<float>(), n, s, o.RenderedLangEmbedding.data_ptr<float>(), current_stream()), "nrf_render_clip_embedding");
		return o;
	}

private:
	/// one pass of the fused path over [n, s] depths: sigma_le -> weights (-> rendered embedding)
	bool ExactCoarseOn() const { return ExactCoarse && Precision == NRF_PREC_F16_SPLIT && nrf_lerf_sigma_exact_available(Mlp.m) != 0; }

	LeRFPassOutputs FusedPass(torch::Tensor rays, torch::Tensor z, torch::Tensor rays_d, bool want_embedding, bool exact = false)
	{
		const int64_t n = z.size(0); const int s = (int)z.size(1); const int stride = (int)rays.size(1);
		auto opt = rays.options();
		auto pts = torch::empty({n * s, 3}, opt);
		check(nrf_points(rays.data_ptr<float>(), stride, z.data_ptr<float>(), n, s, pts.data_ptr<float>(), current_stream()), "nrf_points");
		auto sig = torch::empty({n, s}, opt);
		torch::Tensor x, keep;
		if (LevelMajor) {
			x = torch::empty({16, n * s, 8}, opt.dtype(torch::kFloat16)); keep = torch::empty({n * s}, opt.dtype(torch::kUInt8));
			check(nrf_hash_encode_lm_f16(LangEmbedFn->GetHandle(), pts.data_ptr<float>(), n * s, x.data_ptr(), keep.data_ptr<uint8_t>(), current_stream()), "nrf_hash_encode_lm_f16");
			if (exact) check(nrf_lerf_sigma_exact_lm_strided(Mlp.m, x.data_ptr(), n * s, keep.data_ptr<uint8_t>(), n * s, sig.data_ptr<float>(), nullptr, 0, current_stream()), "nrf_lerf_sigma_exact_lm_strided");
			else check(nrf_lerf_sigma_lm(Mlp.m, x.data_ptr(), keep.data_ptr<uint8_t>(), n * s, sig.data_ptr<float>(), current_stream()), "nrf_lerf_sigma_lm");
		} else {
			torch::Tensor kb;
			std::tie(x, kb) = LangEmbedFn->forward(pts);
			keep = kb.to(torch::kUInt8);
			check(nrf_lerf_sigma(Mlp.m, x.data_ptr<float>(), keep.data_ptr<uint8_t>(), n * s, sig.data_ptr<float>(), current_stream()), "nrf_lerf_sigma");
		}
		LeRFPassOutputs o;
		o.WeightsLE = torch::empty({n, s}, opt); o.DepthMapLE = torch::empty({n}, opt); o.DispMapLE = torch::empty({n}, opt); o.AccMapLE = torch::empty({n}, opt);
		check(nrf_raw2weights(sig.data_ptr<float>(), 1, 0, z.data_ptr<float>(), rays_d.data_ptr<float>(), 3, n, s, o.WeightsLE.data_ptr<float>(), o.DepthMapLE.data_ptr<float>(),
			o.DispMapLE.data_ptr<float>(), o.AccMapLE.data_ptr<float>(), current_stream()), "nrf_raw2weights");
		if (want_embedding) {
			const int E = GetLangEmbedDim();
			auto acc = torch::empty({n, (int64_t)E}, opt);
			if (LevelMajor) check(nrf_lerf_render_embedding_lm(Mlp.m, x.data_ptr(), o.WeightsLE.data_ptr<float>(), n, s, acc.data_ptr<float>(), current_stream()), "nrf_lerf_render_embedding_lm");
			else check(nrf_lerf_render_embedding(Mlp.m, x.data_ptr<float>(), o.WeightsLE.data_ptr<float>(), n, s, acc.data_ptr<float>(), current_stream()), "nrf_lerf_render_embedding");
			auto ones = torch::ones({n, 1}, opt);
			o.RenderedLangEmbedding = torch::empty({n, (int64_t)E}, opt);          // the final normalize of RenderCLIPEmbedding (LeRFRenderer.h:53)
			check(nrf_render_clip_embedding(acc.data_ptr<float>(), E, E, ones.data_ptr<float>(), n, 1, o.RenderedLangEmbedding.data_ptr<float>(), current_stream()), "nrf_render_clip_embedding");
		}
		return o;
	}

	/// Both passes with every sample point encoded ONCE (level-major fused path): the fine pass's S + N_importance depths contain the S coarse ones, whose
	/// features and sigma_le exist -- the hash encode and the sigma net run on the N_importance new samples only, the embedding pass reads every depth's feature
	/// column through the merge map of nrf_fine_depths_merge.  Same kernels on the same inputs: equal to two FusedPass calls bit for bit (weights, maps).
	LeRFPassOutputs FusedPassesReusing(torch::Tensor rays, torch::Tensor z, torch::Tensor rays_d, const int ni, torch::Tensor *z_fine)
	{
		const int64_t n = z.size(0); const int s = (int)z.size(1), sf = s + ni; const int stride = (int)rays.size(1);
		const int64_t cols = n * sf, nc = n * s, nn = n * (int64_t)ni;
		auto opt = rays.options();
		const nrf_hash *h = LangEmbedFn->GetHandle();
		auto x = torch::empty({16, cols, 8}, opt.dtype(torch::kFloat16));
		auto keep = torch::empty({cols}, opt.dtype(torch::kUInt8));
		auto sig = torch::empty({cols}, opt);                                      // [coarse n*s | new n*ni]: the table's column order
		auto pts = torch::empty({nc, 3}, opt);
		check(nrf_points(rays.data_ptr<float>(), stride, z.data_ptr<float>(), n, s, pts.data_ptr<float>(), current_stream()), "nrf_points");
		check(nrf_hash_encode_lm_f16_strided(h, pts.data_ptr<float>(), nc, x.data_ptr(), cols, keep.data_ptr<uint8_t>(), current_stream()), "nrf_hash_encode_lm_f16_strided");
		// split precision: the sigma pass also leaves (sigma, geo32) per column and the embedding pass starts at LE0 from it
		const bool hand_over = Precision == NRF_PREC_F16_SPLIT && HandOverGeo;
		torch::Tensor geo;
		if (hand_over) geo = torch::empty({(int64_t)nrf_lerf_geo_bytes(cols)}, opt.dtype(torch::kUInt8));
		auto sigma_pass = [&](void *xp, uint8_t *kp, int64_t count, float *sp, int64_t col0) {
			if (hand_over) check(nrf_lerf_sigma_geo_lm_strided(Mlp.m, xp, cols, kp, count, sp, static_cast<char *>(geo.data_ptr()) + col0 * 32, cols, current_stream()), "nrf_lerf_sigma_geo_lm_strided");
			else check(nrf_lerf_sigma_lm_strided(Mlp.m, xp, cols, kp, count, sp, current_stream()), "nrf_lerf_sigma_lm_strided");
		};
		// split precision: the COARSE columns' sigma_le in exact fp32 on the matrix cores (its weights choose the fine samples through a discontinuous function, so the
		// sample set is then the fp32 path's own, bit for bit) -- and, with the hand-over, the sigma net's geo output split from the exact values
		if (ExactCoarseOn()) check(nrf_lerf_sigma_exact_lm_strided(Mlp.m, x.data_ptr(), cols, keep.data_ptr<uint8_t>(), nc, sig.data_ptr<float>(), hand_over ? geo.data_ptr() : nullptr, cols,
			current_stream()), "nrf_lerf_sigma_exact_lm_strided");
		else sigma_pass(x.data_ptr(), keep.data_ptr<uint8_t>(), nc, sig.data_ptr<float>(), 0);
		auto weights = [&](const float *sg, torch::Tensor zz, int ss) {
			LeRFPassOutputs o;
			o.WeightsLE = torch::empty({n, (int64_t)ss}, opt); o.DepthMapLE = torch::empty({n}, opt); o.DispMapLE = torch::empty({n}, opt); o.AccMapLE = torch::empty({n}, opt);
			check(nrf_raw2weights(sg, 1, 0, zz.data_ptr<float>(), rays_d.data_ptr<float>(), 3, n, ss, o.WeightsLE.data_ptr<float>(), o.DepthMapLE.data_ptr<float>(),
				o.DispMapLE.data_ptr<float>(), o.AccMapLE.data_ptr<float>(), current_stream()), "nrf_raw2weights");
			return o;
		};
		LeRFPassOutputs coarse = weights(sig.data_ptr<float>(), z, s);
		auto u = torch::linspace(0.f, 1.f, ni, torch::kFloat).to(rays.device());
		auto zf = torch::empty({n, (int64_t)sf}, opt), z_new = torch::empty({n, (int64_t)ni}, opt);
		auto src = torch::empty({n, (int64_t)sf}, opt.dtype(torch::kInt32));
		check(nrf_fine_depths_merge(z.data_ptr<float>(), coarse.WeightsLE.data_ptr<float>(), n, s, u.data_ptr<float>(), ni, 8, zf.data_ptr<float>(), src.data_ptr<int32_t>(),
			z_new.data_ptr<float>(), current_stream()), "nrf_fine_depths_merge");
		auto pts_new = torch::empty({nn, 3}, opt);
		check(nrf_points(rays.data_ptr<float>(), stride, z_new.data_ptr<float>(), n, ni, pts_new.data_ptr<float>(), current_stream()), "nrf_points");
		void *x_new = static_cast<char *>(x.data_ptr()) + nc * 8 * 2;             // column n*s of level 0
		check(nrf_hash_encode_lm_f16_strided(h, pts_new.data_ptr<float>(), nn, x_new, cols, keep.data_ptr<uint8_t>() + nc, current_stream()), "nrf_hash_encode_lm_f16_strided");
		sigma_pass(x_new, keep.data_ptr<uint8_t>() + nc, nn, sig.data_ptr<float>() + nc, nc);
		// sigma_le stays in column order: the compositing kernel reads the sorted depths' values through the merge map
		LeRFPassOutputs o;
		o.WeightsLE = torch::empty({n, (int64_t)sf}, opt); o.DepthMapLE = torch::empty({n}, opt); o.DispMapLE = torch::empty({n}, opt); o.AccMapLE = torch::empty({n}, opt);
		check(nrf_raw2weights_gather(sig.data_ptr<float>(), 1, 0, src.data_ptr<int32_t>(), zf.data_ptr<float>(), rays_d.data_ptr<float>(), 3, n, sf, o.WeightsLE.data_ptr<float>(),
			o.DepthMapLE.data_ptr<float>(), o.DispMapLE.data_ptr<float>(), o.AccMapLE.data_ptr<float>(), current_stream()), "nrf_raw2weights_gather");
		const int E = GetLangEmbedDim();
		auto acc = torch::empty({n, (int64_t)E}, opt);
		if (hand_over) check(nrf_lerf_render_embedding_lm_geo(Mlp.m, x.data_ptr(), cols, src.data_ptr<int32_t>(), geo.data_ptr(), cols, o.WeightsLE.data_ptr<float>(), n, sf, acc.data_ptr<float>(),
			current_stream()), "nrf_lerf_render_embedding_lm_geo");
		else check(nrf_lerf_render_embedding_lm_gather(Mlp.m, x.data_ptr(), cols, src.data_ptr<int32_t>(), o.WeightsLE.data_ptr<float>(), n, sf, acc.data_ptr<float>(), current_stream()),
			"nrf_lerf_render_embedding_lm_gather");
		auto ones = torch::ones({n, 1}, opt);
		o.RenderedLangEmbedding = torch::empty({n, (int64_t)E}, opt);
		check(nrf_render_clip_embedding(acc.data_ptr<float>(), E, E, ones.data_ptr<float>(), n, 1, o.RenderedLangEmbedding.data_ptr<float>(), current_stream()), "nrf_render_clip_embedding");
		if (z_fine) *z_fine = zf;
		return o;
	}

public:
	bool ReuseFeatures = true;          ///< level-major fused path: encode every sample point once per render (false: two plain passes)
	bool HandOverGeo = true;            ///< split precision, reuse path: the embedding pass takes the sigma net's output from the sigma pass (false: re-evaluates it)
	bool ExactCoarse = true;            ///< split precision: the coarse pass's sigma_le in exact fp32 on the matrix cores (sigma_lerf_f32.hip): the fp32 path's sample set

	/// LeRFRenderer::RenderRays (LeRFRenderer.cpp:85-187), deterministic path (Perturb = 0, RawNoiseStd = 0, ThinRay): the fused matrix-core passes when the
	/// sample counts are multiples of 32 (a wave's 32-point tile lies inside one ray), the fp32 stage path otherwise.  z_fine (optional) receives the fine depths.
	LeRFPassOutputs RenderRays(torch::Tensor ray_batch, const int n_samples, const bool lin_disp = false, const int n_importance = 0, const bool return_weights = true,
		torch::Tensor *z_fine = nullptr)
	{
		auto rays = dev_f32(ray_batch);
		const int64_t n = rays.size(0); const int stride = (int)rays.size(1);
		const int s = n_samples, ni = n_importance;
		using torch::indexing::Slice;
		auto opt = rays.options();
		auto t = torch::linspace(0.f, 1.f, s, torch::kFloat).to(rays.device());
		auto z = torch::empty({n, s}, opt);
		check(nrf_z_vals(rays.data_ptr<float>(), stride, n, t.data_ptr<float>(), s, lin_disp, z.data_ptr<float>(), current_stream()), "nrf_z_vals");
		auto rays_d = rays.index({Slice(), Slice(3, 6)}).contiguous();
		if (SingleCallOk(s, ni) && n * (int64_t)(s + ni) < ((int64_t)1 << 31)) {
			// the whole chunk as ONE library call (nrf_lerf_render_rays): no torch ops, Relevancy included when prompts are set
			LeRFPassOutputs out; nrf_lerf_outputs ro{};
			alloc_outputs(n, s + ni, return_weights, opt, out, ro, z_fine);
			nrf_render_params p = pass_params(s, ni, lin_disp);
			const size_t wsb = nrf_lerf_render_rays_workspace_bytes(Pass, n, &p);
			check(nrf_lerf_render_rays(Pass, rays.data_ptr<float>(), stride, n, &p, linspace01(s, rays.device()).data_ptr<float>(), linspace01(ni, rays.device()).data_ptr<float>(), &ro,
				workspace(wsb, rays.device()), wsb, current_stream()), "nrf_lerf_render_rays");
			return out;
		}
		const bool fused = Fused && s % 32 == 0 && (ni == 0 || (s + ni) % 32 == 0);
		auto stage = [&](torch::Tensor zz) {
			auto pts = torch::empty({n, zz.size(1), 3}, opt);
			check(nrf_points(rays.data_ptr<float>(), stride, zz.data_ptr<float>(), n, (int)zz.size(1), pts.data_ptr<float>(), current_stream()), "nrf_points");
			return RawToLEOutputs(RunLENetwork(pts), zz, rays_d, GetLangEmbedDim());
		};
		if (fused && LevelMajor && ReuseFeatures && ni > 0 && n * (int64_t)(s + ni) < ((int64_t)1 << 31)) {
			LeRFPassOutputs out = FusedPassesReusing(rays, z, rays_d, ni, z_fine);
			if (Positives.defined() && Negatives.defined()) out.Relevancy = Relevancy(out.RenderedLangEmbedding, Positives, Negatives);      // LeRFRenderer.cpp:79
			if (!return_weights) { out.WeightsLE = torch::Tensor(); out.LangEmbedding = torch::Tensor(); out.RenderedLangEmbedding = torch::Tensor(); }
			return out;
		}
		LeRFPassOutputs out = fused ? FusedPass(rays, z, rays_d, ni == 0, ni > 0 && LevelMajor && ExactCoarseOn()) : stage(z);
		if (ni > 0) {
			auto u = torch::linspace(0.f, 1.f, ni, torch::kFloat).to(rays.device());
			auto zf = torch::empty({n, (int64_t)(s + ni)}, opt);
			check(nrf_fine_depths(z.data_ptr<float>(), out.WeightsLE.data_ptr<float>(), n, s, u.data_ptr<float>(), ni, 8, zf.data_ptr<float>(), current_stream()), "nrf_fine_depths");
			out = fused ? FusedPass(rays, zf, rays_d, true) : stage(zf);
			if (z_fine) *z_fine = zf;
		}
		if (Positives.defined() && Negatives.defined() && out.RenderedLangEmbedding.defined()) out.Relevancy = Relevancy(out.RenderedLangEmbedding, Positives, Negatives);      // LeRFRenderer.cpp:79
		if (!return_weights) { out.WeightsLE = torch::Tensor(); out.LangEmbedding = torch::Tensor(); out.RenderedLangEmbedding = torch::Tensor(); }      // LeRFRenderer.cpp:180-185
		return out;
	}

	/// LeRFRenderer::Render (LeRFRenderer.cpp:265-330) for a pose or an explicit ray batch: rays, AABB clipping, the chunk loop, Near / Far.  row0 / rows:
	/// a row tile of the frame (multi-GPU sharding).  Returns the concatenated per-ray outputs (the reference does not reshape them either).
	LeRFPassOutputs Render(const int h, const int w, torch::Tensor k, torch::Tensor bounding_box, const int n_samples, const int n_importance, const int chunk,
		torch::Tensor c2w, const bool use_viewdirs = true, const bool lin_disp = false, const bool return_weights = true, float *near_out = nullptr, float *far_out = nullptr,
		int row0 = 0, int rows = -1)
	{
		if (rows < 0) rows = h - row0;
		const auto dev = torch::Device(torch::kCUDA, c10::hip::getCurrentHIPStream().device_index());
		auto K = host_floats(k), M = host_floats(c2w.index({torch::indexing::Slice(torch::indexing::None, 3), torch::indexing::Slice(torch::indexing::None, 4)}));
		if (SingleCallOk(n_samples, n_importance)) {
			// LeRFRenderer::Render as ONE library call (nrf_lerf_render_rows): rays, AABB clipping, the Chunk loop on the library's lanes, Relevancy, the tile's Near / Far
			nrf_view v{};
			v.h = h; v.w = w; v.row0 = row0; v.rows = rows; v.use_viewdirs = use_viewdirs; v.ndc = 0; v.chunk = chunk;
			for (int i = 0; i < 9; i++) v.K[i] = K[i];
			for (int i = 0; i < 12; i++) v.c2w[i] = M[i];
			auto bbv = host_floats(bounding_box);
			TORCH_CHECK(bbv.size() == 6, "Render: bounding_box must hold [min xyz, max xyz]");
			for (int i = 0; i < 6; i++) v.bbox[i] = bbv[i];
			const auto opt = torch::TensorOptions().dtype(torch::kFloat32).device(dev);
			const int64_t nr = (int64_t)rows * w;
			LeRFPassOutputs out; nrf_lerf_outputs ro{};
			alloc_outputs(nr, n_samples + n_importance, return_weights, opt, out, ro, nullptr);
			nrf_render_params p = pass_params(n_samples, n_importance, lin_disp);
			auto nf = torch::empty({2}, opt);
			const size_t wsb = nrf_lerf_render_rows_workspace_bytes(Pass, &v, &p);
			check(nrf_lerf_render_rows(Pass, &v, &p, linspace01(n_samples, dev).data_ptr<float>(), linspace01(n_importance, dev).data_ptr<float>(), &ro, nullptr, nf.data_ptr<float>(),
				workspace(wsb, dev), wsb, current_stream()), "nrf_lerf_render_rows");
			if (near_out || far_out) { auto nfh = nf.cpu(); if (near_out) *near_out = nfh[0].item<float>(); if (far_out) *far_out = nfh[1].item<float>(); }
			return out;
		}
		auto o = torch::empty({(int64_t)rows * w, 3}, torch::TensorOptions().dtype(torch::kFloat32).device(dev)), d = torch::empty_like(o);
		check(nrf_get_rays(h, w, K.data(), M.data(), row0, rows, o.data_ptr<float>(), d.data_ptr<float>(), nullptr, current_stream()), "nrf_get_rays");
		const int64_t n = o.size(0);
		const int stride = use_viewdirs ? 11 : 8;
		auto bb = host_floats(bounding_box);
		auto rays_ = torch::empty({n, stride}, o.options());
		check(nrf_pack_rays(o.data_ptr<float>(), d.data_ptr<float>(), bb.data(), n, use_viewdirs, rays_.data_ptr<float>(), current_stream()), "nrf_pack_rays");
		std::vector<torch::Tensor> e, w_, dep, disp, acc, rel;
		for (int64_t i = 0; i < n; i += chunk) {
			auto part = RenderRays(rays_.index({torch::indexing::Slice(i, std::min<int64_t>(i + chunk, n))}), n_samples, lin_disp, n_importance, return_weights);
			if (part.RenderedLangEmbedding.defined()) e.push_back(part.RenderedLangEmbedding);
			if (part.WeightsLE.defined()) w_.push_back(part.WeightsLE);
			if (part.Relevancy.defined()) rel.push_back(part.Relevancy);
			dep.push_back(part.DepthMapLE); disp.push_back(part.DispMapLE); acc.push_back(part.AccMapLE);
		}
		LeRFPassOutputs out;
		if (!e.empty()) out.RenderedLangEmbedding = torch::cat(e, 0);
		if (!w_.empty()) out.WeightsLE = torch::cat(w_, 0);
		if (!rel.empty()) out.Relevancy = torch::cat(rel, 0);
		out.DepthMapLE = torch::cat(dep, 0); out.DispMapLE = torch::cat(disp, 0); out.AccMapLE = torch::cat(acc, 0);
		float nr = 0.f, fr = 0.f;
		check(nrf_near_far_range(rays_.data_ptr<float>(), n, stride, &nr, &fr, current_stream()), "nrf_near_far_range");
		if (near_out) *near_out = nr;
		if (far_out) *far_out = fr;
		return out;
	}
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
	nrf_mlp_small_desc Small{};   ///the NeRFSmall description given to SyncWeights (training path: nrf_mlp_backward is built for this family)
	bool HasSmall = false;
	uint64_t MlpSignature = 0;    ///ATen version counters + storage addresses of the network's parameters at the last upload
	uint64_t TrainCalls = 0;      ///training renders so far: every one draws afresh from the counter RNG, as the reference does from torch's global generator
	torch::Tensor TrainWorkspace, HashTrainWorkspace;   ///the fused backward's operand slots + overflow word, the binned scatter's records (kept apart from the render workspace)

	void *workspace(size_t bytes, torch::Device dev)
	{
		if (!Workspace.defined() || (size_t)Workspace.numel() < bytes) Workspace = torch::empty({(int64_t)bytes}, torch::TensorOptions().dtype(torch::kUInt8).device(dev));
		return Workspace.data_ptr();
	}
public:
	/// small = {num_layers, hidden_dim, geo_feat_dim, num_layers_color, hidden_dim_color} of the NeRFSmall the executor built
	/// (NeRFExecutor.h:479-493), or nerf = {depth, width, output_ch, skip, use_viewdirs} for the classic NeRF.
	HipNeRFRenderer(TEmbedder embed_fn, TEmbedDirs embeddirs_fn, TNeRF nerf, int precision = NRF_PREC_F16_MFMA) : Base(embed_fn, embeddirs_fn, nerf), Precision(precision)
	{
		nrfpp::live_renderers().add(this);
	}
	~HipNeRFRenderer() override { nrfpp::live_renderers().remove(this); nrf_renderer_destroy(Renderer); }

	void SetPrecision(int precision) { Precision = precision; }
	/// arithmetic of the training backward: -1 (default) follows the render precision -- the fused fp16 matrix-core chain unless the renderer is in NRF_PREC_F32;
	/// 0: always the fp32 layer kernels (the reference-pinned parity path); 1: the fp16 chain whenever the network is of the fused kernel's family
	int TrainBackwardArithmetic = -1;
	bool ReuseRenderFeatures = true;                    ///the fused backward reads the hash features the forward render left in the workspace (nrf_renderer_last_features) when it can
	bool ReusedRenderFeatures = false;                  ///... and whether the last backward did
	int64_t F16BackwardOverflows = 0;                   ///steps whose fp16 chain reported a non-finite value and were redone in fp32
	bool WantsF16Backward() const { return TrainBackwardArithmetic == 1 || (TrainBackwardArithmetic == -1 && Precision != NRF_PREC_F32); }
	void SetSeed(uint64_t seed) { Seed = seed; }
	/// What a matrix-core render does when its network outputs hold a NaN / Inf (nrf_render_params.overflow_policy; no reference counterpart: NeRFRenderParams is the
	/// reference's struct, so the policy is a property of the renderer).  NRF_OVERFLOW_AUTO (default): the chunk is rendered again in NRF_PREC_F32 before Render returns;
	/// NRF_OVERFLOW_ERROR: Render throws (NRF_ERR_NONFINITE); NRF_OVERFLOW_DEFERRED: no wait in the frame loop, ask Nonfinite(); NRF_OVERFLOW_IGNORE.
	int OverflowPolicy = NRF_OVERFLOW_AUTO;
	void SetOverflowPolicy(int policy) { OverflowPolicy = policy; }
	/// {chunks whose non-finite word was set, chunks rendered again in fp32} since the renderer was built (settles the deferred words that have arrived)
	std::pair<int64_t, int64_t> Nonfinite() const
	{
		int64_t flagged = 0, redone = 0;
		nrfpp::check(nrf_renderer_nonfinite(Renderer, &flagged, &redone), "nrf_renderer_nonfinite");
		return {flagged, redone};
	}

	/// (Re)read the network's parameters and rebuild the device-side images; call after construction, load or an optimizer step.
	void SyncWeights(const nrf_mlp_small_desc *small, const nrf_mlp_nerf_desc *classic)
	{
		auto blob = parameter_blob(this->NeRF);
		nrf_mlp *m = nullptr;
		if (small) { TORCH_CHECK((int64_t)blob.size() == nrf_mlp_small_param_count(small), "NeRFSmall parameter count mismatch"); check(nrf_mlp_small_create(small, blob.data(), 0, current_stream(), &m), "nrf_mlp_small_create"); }
		else { TORCH_CHECK((int64_t)blob.size() == nrf_mlp_nerf_param_count(classic), "NeRF parameter count mismatch"); check(nrf_mlp_nerf_create(classic, blob.data(), 0, current_stream(), &m), "nrf_mlp_nerf_create"); }
		Mlp.reset(m);
		HasSmall = small != nullptr;
		if (small) Small = *small;
		MlpSignature = MlpSignatureOf();
		nrf_renderer_destroy(Renderer); Renderer = nullptr;
		nrf_renderer_desc d{};
		if constexpr (std::is_same_v<TEmbedder, HipHashEmbedder>) { this->EmbedFn->Sync(); d.hash = this->EmbedFn->GetHandle(); }
		else { d.hash = nullptr; d.pe_freqs = this->EmbedFn->GetMultires(); }
		if constexpr (std::is_same_v<TEmbedDirs, HipSHEncoder>) { d.dirs_encoder = this->EmbeddirsFn->GetVariant() == NRF_SH_CUDA ? NRF_DIRS_SH_CUDA : NRF_DIRS_SH_LIBTORCH; d.dirs_param = this->EmbeddirsFn->GetDegree(); }
		else { d.dirs_encoder = NRF_DIRS_PE; d.dirs_param = this->EmbeddirsFn->GetMultires(); }
		d.mlp = Mlp.m;
		check(nrf_renderer_create(&d, &Renderer), "nrf_renderer_create");
	}

	uint64_t MlpSignatureOf()
	{
		uint64_t sig = 1469598103934665603ull;
		for (auto &p : this->NeRF->parameters()) { sig = (sig ^ (uint64_t)p._version()) * 1099511628211ull; sig = (sig ^ (uint64_t)(uintptr_t)p.data_ptr()) * 1099511628211ull; }
		return sig;
	}
	/// the network's parameters as one blob in named_parameters() order (== the blob order of nerfpp_hip.h), still attached to them: a gradient w.r.t. this tensor
	/// reaches every Linear weight through torch::cat's backward
	torch::Tensor BlobForGrad()
	{
		std::vector<torch::Tensor> flat;
		for (auto &p : this->NeRF->named_parameters()) flat.push_back(p.value().reshape({-1}));
		return torch::cat(flat, 0);
	}
	/// Upload the network's parameters when they changed since the last upload (optimizer step, torch::load, copy_): the host never has to call SyncWeights again
	/// after construction -- NeRFExecutor::Train steps its optimizer and renders test views without knowing about this class (NeRFExecutor.h:985, :1007-1042).
	void SyncIfChanged()
	{
		TORCH_CHECK(Renderer != nullptr, "HipNeRFRenderer: call SyncWeights(small | classic description) once after construction");
		const uint64_t sig = MlpSignatureOf();
		if (sig != MlpSignature) {
			PhaseScope ps("mlp.sync_params");
			torch::NoGradGuard ng;
			torch::Tensor blob = BlobForGrad().to(torch::kFloat32).contiguous();
			check(nrf_mlp_set_params(Mlp.m, blob.data_ptr<float>(), blob.is_cuda() ? 1 : 0, current_stream()), "nrf_mlp_set_params");
			MlpSignature = sig;
		}
		if constexpr (std::is_same_v<TEmbedder, HipHashEmbedder>) this->EmbedFn->SyncIfChanged();
	}
	bool WantsGrad()
	{
		if (!torch::GradMode::is_enabled()) return false;
		for (auto &p : this->NeRF->parameters()) if (p.requires_grad()) return true;
		if constexpr (std::is_same_v<TEmbedder, HipHashEmbedder>) return this->EmbedFn->AnyRequiresGrad();
		return false;
	}

	// ---- the training render: NeRFRenderer::Render on a ray batch as ONE autograd node (NeRFExecutor.h:876-923: Render -> huber_loss -> loss.backward()) ----
	struct TrainState {           // what the backward needs besides the saved tensors
		HipNeRFRenderer *self; int s; bool fine; bool white_bkgr; float noise_std, precond_alpha, cone_angle; bool has_cone; uint64_t seed; std::vector<float> bbox;
		int64_t view_serial = -1;          // nrf_renderer_last_features' serial right after the forward's render (-1: it left no feature view)
	};
	/// forward: rays_ [n, 8 | 11] -> {rgb, disp, acc, depth, weights | empty, raw}; gradients flow to `table` and `blob` from d loss / d rgb only (the fine pass:
	/// z_samples are detached, NeRFRenderer.h:429; disparity / accumulation / depth / weights / raw are marked non-differentiable -- the reference's loss reads RGBMap, :882-887)
	struct RenderFn : public torch::autograd::Function<RenderFn> {
		static torch::autograd::variable_list forward(torch::autograd::AutogradContext *ctx, torch::Tensor rays_, torch::Tensor table, torch::Tensor blob, int64_t self_i, int64_t rp_i,
			double cone_value)          // < 0: thin rays
		{
			PhaseScope ps("train.forward");
			torch::Tensor cone_angle; if (cone_value >= 0.0) cone_angle = torch::tensor((float)cone_value);
			auto *self = reinterpret_cast<HipNeRFRenderer *>(self_i);
			const NeRFRenderParams &rp = *reinterpret_cast<const NeRFRenderParams *>(rp_i);
			const int64_t n = rays_.size(0); const int stride = (int)rays_.size(1);
			const int s = rp.NSamples, ni = rp.NImportance, so = ni > 0 ? s + ni : s;
			const auto opt = rays_.options();
			auto rgb = torch::empty({n, 3}, opt), disp = torch::empty({n}, opt), acc = torch::empty({n}, opt), depth = torch::empty({n}, opt);
			auto weights = rp.ReturnWeights ? torch::empty({n, (int64_t)so}, opt) : torch::Tensor();
			auto raw = torch::empty({n, (int64_t)so, 4}, opt), z = torch::empty({n, (int64_t)so}, opt);
			nrf_render_outputs ro{};
			ro.d_rgb = rgb.data_ptr<float>(); ro.d_disp = disp.data_ptr<float>(); ro.d_acc = acc.data_ptr<float>(); ro.d_depth = depth.data_ptr<float>();
			ro.d_weights = rp.ReturnWeights ? weights.data_ptr<float>() : nullptr;
			ro.d_raw = raw.data_ptr<float>();
			if (ni > 0) ro.d_z_fine = z.data_ptr<float>(); else ro.d_z_coarse = z.data_ptr<float>();
			nrf_render_params p = self->make_params(rp, cone_angle, 0);
			p.seed = self->Seed + 0x9E3779B97F4A7C15ull * self->TrainCalls++;
			torch::Tensor t = self->linspace01(s, rays_.device());
			torch::Tensor u; if (ni > 0) u = self->linspace01(ni, rays_.device());
			if (n > 0) {
				const size_t wsb = nrf_batchify_rays_workspace_bytes(self->Renderer, n, rp.Chunk, &p);
				check(nrf_batchify_rays(self->Renderer, rays_.data_ptr<float>(), stride, n, rp.Chunk, &p, t.data_ptr<float>(), u.defined() ? u.data_ptr<float>() : nullptr, &ro,
					self->workspace(wsb, rays_.device()), wsb, current_stream()), "nrf_batchify_rays");
			}
			{
				// the hash features the render has just left in the workspace (a single-chunk render of the feature-reusing fast path): the backward reads them instead of
				// encoding the fine points again, provided the renderer has rendered nothing in between (same serial)
				const void *f = nullptr; const uint8_t *k = nullptr; const int32_t *sr = nullptr; int64_t cols = 0, vn = 0; int vsf = 0; uint64_t serial = 0;
				const int rcv = n > 0 ? nrf_renderer_last_features(self->Renderer, &f, &cols, &k, &sr, &vn, &vsf, &serial) : NRF_ERR_UNSUPPORTED;
				ctx->saved_data["view_serial"] = (rcv == NRF_OK && vn == n && vsf == so) ? (int64_t)serial : (int64_t)-1;
			}
			ctx->save_for_backward({rays_, raw, z});
			// the backward's state as plain values in the node itself (no heap object to leak when the graph is dropped without a backward, nothing consumed by a backward:
			// retain_graph works); the renderer is looked up in the registry of live ones, so a backward after its destruction fails loudly instead of dereferencing it
			ctx->saved_data["self"] = self_i;
			ctx->saved_data["s"] = (int64_t)so; ctx->saved_data["fine"] = ni > 0; ctx->saved_data["white_bkgr"] = (bool)rp.WhiteBkgr;
			ctx->saved_data["noise_std"] = (double)rp.RawNoiseStd; ctx->saved_data["precond_alpha"] = (double)(ni > 0 ? rp.StochasticPreconditioningAlpha : 0.f);
			ctx->saved_data["cone_angle"] = (double)p.cone_angle; ctx->saved_data["has_cone"] = p.has_cone != 0;
			ctx->saved_data["seed"] = (int64_t)p.seed;
			ctx->saved_data["bbox"] = p.has_bbox ? std::vector<double>(p.bbox, p.bbox + 6) : std::vector<double>();
			ctx->saved_data["table_sizes"] = table.sizes().vec();
			ctx->saved_data["blob_numel"] = blob.numel();
			std::vector<torch::Tensor> nd{disp, acc, depth, raw};
			if (weights.defined()) nd.push_back(weights);
			ctx->mark_non_differentiable(nd);
			return {rgb, disp, acc, depth, weights.defined() ? weights : torch::empty({0}, opt), raw};
		}
		static torch::autograd::variable_list backward(torch::autograd::AutogradContext *ctx, torch::autograd::variable_list grads)
		{
			auto &sd = ctx->saved_data;
			const int64_t self_i = sd["self"].toInt();
			TORCH_CHECK(self_i != 0 && nrfpp::live_renderers().alive(reinterpret_cast<const void *>(self_i)),
				"HipNeRFRenderer: backward through a Render() whose renderer has been destroyed");
			TrainState st{reinterpret_cast<HipNeRFRenderer *>(self_i), (int)sd["s"].toInt(), sd["fine"].toBool(), sd["white_bkgr"].toBool(), (float)sd["noise_std"].toDouble(),
				(float)sd["precond_alpha"].toDouble(), (float)sd["cone_angle"].toDouble(), sd["has_cone"].toBool(), (uint64_t)sd["seed"].toInt(), {}, sd["view_serial"].toInt()};
			for (double v : sd["bbox"].toDoubleVector()) st.bbox.push_back((float)v);
			auto saved = ctx->get_saved_variables();
			auto [g_table, g_blob] = st.self->TrainBackward(st, saved[0], saved[1], saved[2], grads[0], sd["table_sizes"].toIntVector(), sd["blob_numel"].toInt());
			return {torch::Tensor(), g_table, g_blob, torch::Tensor(), torch::Tensor(), torch::Tensor()};
		}
	};

	torch::Tensor rng_fill(uint64_t seed, uint32_t stream, int64_t count, bool normal, torch::TensorOptions opt)
	{
		auto out = torch::empty({count}, opt);
		check(nrf_rng_fill(seed, stream, 0, count, normal, out.data_ptr<float>(), current_stream()), "nrf_rng_fill");
		return out;
	}

	/// d loss / d RGBMap -> (d loss / d table, d loss / d blob): RawToOutputs' backward (TruncExp's clamp included) -> the keep mask -> the network (fp32 layer-wise
	/// kernels, pinned to the reference's autograd by golden `train_hash`) -> the hash grid.  The sample points of the forward are re-formed from the saved depths; the
	/// draws of its stochastic branches (cone rays, preconditioning, raw noise) are regenerated from the same (seed, stream, index) of the counter RNG.
	std::pair<torch::Tensor, torch::Tensor> TrainBackward(const TrainState &st, torch::Tensor rays, torch::Tensor raw, torch::Tensor z, torch::Tensor g_rgb_in,
		std::vector<int64_t> table_sizes, int64_t blob_numel)
	{
		PhaseScope ps("train.backward");
		const int64_t n = rays.size(0); const int stride = (int)rays.size(1), s = st.s;
		const auto opt = rays.options();
		auto g_table = torch::zeros(table_sizes, opt), g_blob = torch::zeros({blob_numel}, opt);
		if (n == 0 || !g_rgb_in.defined()) return {g_table, g_blob};
		if constexpr (!std::is_same_v<TEmbedder, HipHashEmbedder>) {
			// the classic configuration (HipEmbedder / HipEmbedder / NeRFImpl): the positional encodings carry no parameters, the chain ends at the network's gradient
			// (nrf_mlp_backward on NeRFImpl: biases, skip concat, view-direction head -- pinned by the reference's autograd, goldens mlp_nerf_bwd*, train_classic)
			TORCH_CHECK(st.precond_alpha == 0.f && !st.has_cone, "training the classic model through HipNeRFRenderer: thin rays without stochastic preconditioning");
			auto g_rgb = dev_f32(g_rgb_in).reshape({n, 3});
			auto g_raw = torch::empty_like(raw);
			torch::Tensor noise;
			if (st.noise_std > 0.f) noise = rng_fill(st.seed, st.fine ? NRF_RNG_NOISE_FINE : NRF_RNG_NOISE_COARSE, n * s, true, opt);
			check(nrf_raw2outputs_backward_noise(raw.data_ptr<float>(), z.data_ptr<float>(), rays.data_ptr<float>() + 3, stride, n, s, 4, st.white_bkgr,
				noise.defined() ? noise.data_ptr<float>() : nullptr, st.noise_std, g_rgb.data_ptr<float>(), g_raw.data_ptr<float>(), current_stream()), "nrf_raw2outputs_backward");
			auto pts = torch::empty({n * s, 3}, opt);
			check(nrf_points(rays.data_ptr<float>(), stride, z.data_ptr<float>(), n, s, pts.data_ptr<float>(), current_stream()), "nrf_points");
			torch::Tensor x = this->EmbedFn->forward(pts).first;
			if (stride == 11) {
				using torch::indexing::Slice;
				torch::Tensor dirs = this->EmbeddirsFn->forward(rays.index({Slice(), Slice(8, 11)}).contiguous()).first;
				x = torch::cat({x, dirs.unsqueeze(1).expand({n, (int64_t)s, dirs.size(1)}).reshape({n * s, dirs.size(1)})}, 1).contiguous();
			}
			const size_t wsb = nrf_mlp_backward_workspace_bytes(Mlp.m, n * s);
			check(nrf_mlp_backward(Mlp.m, x.data_ptr<float>(), g_raw.data_ptr<float>(), n * s, g_blob.data_ptr<float>(), nullptr, workspace(wsb, rays.device()), wsb,
				current_stream()), "nrf_mlp_backward");
		}
		else {
			auto g_rgb = dev_f32(g_rgb_in).reshape({n, 3});
			auto g_raw = torch::empty_like(raw);
			torch::Tensor noise;
			if (st.noise_std > 0.f) noise = rng_fill(st.seed, st.fine ? NRF_RNG_NOISE_FINE : NRF_RNG_NOISE_COARSE, n * s, true, opt);
			check(nrf_raw2outputs_backward_noise(raw.data_ptr<float>(), z.data_ptr<float>(), rays.data_ptr<float>() + 3, stride, n, s, 4, st.white_bkgr,
				noise.defined() ? noise.data_ptr<float>() : nullptr, st.noise_std, g_rgb.data_ptr<float>(), g_raw.data_ptr<float>(), current_stream()), "nrf_raw2outputs_backward");
			auto pts = torch::empty({n * s, 3}, opt);
			check(nrf_points(rays.data_ptr<float>(), stride, z.data_ptr<float>(), n, s, pts.data_ptr<float>(), current_stream()), "nrf_points");
			const float *bb = st.bbox.size() == 6 ? st.bbox.data() : nullptr;
			if (st.precond_alpha > 0.f) {          // NeRFRenderer.h:433-443 (fine pass only)
				TORCH_CHECK(bb, "stochastic preconditioning needs the bounding box");
				torch::Tensor pn = rng_fill(st.seed, NRF_RNG_PRECOND, n * s * 3, true, opt);
				auto out = torch::empty_like(pts);
				check(nrf_precondition(pts.data_ptr<float>(), pn.data_ptr<float>(), st.precond_alpha, bb, n * s, out.data_ptr<float>(), current_stream()), "nrf_precondition");
				pts = out;
			}
			if (st.has_cone) {                     // TangentScatter, NeRFRenderer.h:307-362
				torch::Tensor ur = rng_fill(st.seed, st.fine ? NRF_RNG_R_FINE : NRF_RNG_R_COARSE, n * s, false, opt), ut = rng_fill(st.seed, st.fine ? NRF_RNG_THETA_FINE : NRF_RNG_THETA_COARSE, n * s, false, opt);
				auto out = torch::empty_like(pts);
				check(nrf_tangent_scatter(pts.data_ptr<float>(), rays.data_ptr<float>(), stride, z.data_ptr<float>(), n, s, st.cone_angle, ur.data_ptr<float>(), ut.data_ptr<float>(), bb,
					out.data_ptr<float>(), current_stream()), "nrf_tangent_scatter");
				pts = out;
			}
			const nrf_hash *h = this->EmbedFn->GetHandle();
			const int in_ch = this->EmbedFn->GetOutputDims();
			using torch::indexing::Slice;
			torch::Tensor dirs;                    // UseViewdirs: the direction encoding of each ray (repeated for its samples, NeRFRenderer.h:179-181)
			if (stride == 11) dirs = this->EmbeddirsFn->forward(rays.index({Slice(), Slice(8, 11)}).contiguous()).first;
			auto keep = torch::empty({n * s}, opt.dtype(torch::kUInt8));
			auto g_x = torch::empty({n * s, (int64_t)in_ch}, opt);
			torch::Tensor emb;
			auto fp32_input = [&]() -> torch::Tensor {              // x [p, in_ch + dirs] fp32 rows, as RunNetwork forms them (:175-184)
				if (!emb.defined()) {
					emb = torch::empty({n * s, (int64_t)in_ch}, opt);
					check(nrf_hash_encode(h, pts.data_ptr<float>(), n * s, emb.data_ptr<float>(), keep.data_ptr<uint8_t>(), current_stream()), "nrf_hash_encode");
					check(nrf_mask_sigma_grad(keep.data_ptr<uint8_t>(), n * s, 4, g_raw.data_ptr<float>(), current_stream()), "nrf_mask_sigma_grad");
				}
				if (!dirs.defined()) return emb;
				return torch::cat({emb, dirs.unsqueeze(1).expand({n, (int64_t)s, dirs.size(1)}).reshape({n * s, dirs.size(1)})}, 1).contiguous();
			};
			// The gradient chain of the network.  With the renderer in a matrix-core precision: ONE fused matrix-core kernel (fp16 operands, fp32 accumulation, loss scale chosen on
			// the device from max |g_raw|; mlp_small_bwd_mfma.hip) fed, where the grid is the CuHashEmbedder L16 F2 one, from the level-major fp16 features of the render fast
			// path -- what nerfpp_amd/train.py's Trainer(mlp_backward="f16", hash_backward="binned") issues.  Its overflow word is read back (one host synchronisation per step,
			// as a loss scaler costs): a step whose fp16 chain met a non-finite value is redone by the fp32 layer kernels below instead of being skipped.  With the renderer in
			// NRF_PREC_F32 (the parity mode): the fp32 layer kernels, pinned to the reference's autograd (golden train_hash).
			bool f16_done = false;
			if (WantsF16Backward()) {
				PhaseScope pf("bwd.f16_chain");
				const size_t wsb = nrf_mlp_backward_f16_workspace_bytes(Mlp.m, n * s);
				if (!TrainWorkspace.defined() || (size_t)TrainWorkspace.numel() < wsb) TrainWorkspace = torch::empty({(int64_t)wsb}, opt.dtype(torch::kUInt8));
				int rc;
				const bool lm = this->EmbedFn->Mode == NRF_HASH_CU && this->EmbedFn->NLevels == 16 && this->EmbedFn->NFeaturesPerLevel == 2 && dirs.defined() && dirs.size(1) == 16;
				if (lm) {
					auto dirs16 = dirs.to(torch::kFloat16).contiguous();
					// the forward render's own features where they are still in the workspace (TrainState::view_serial), else a second encode of the fine points
					const void *vf = nullptr; const uint8_t *vk = nullptr; const int32_t *vs = nullptr; int64_t vcols = 0, vn = 0; int vsf = 0; uint64_t serial = 0;
					const bool view = ReuseRenderFeatures && st.view_serial >= 0 && st.fine && !st.has_cone && st.precond_alpha == 0.f &&
						nrf_renderer_last_features(Renderer, &vf, &vcols, &vk, &vs, &vn, &vsf, &serial) == NRF_OK && (int64_t)serial == st.view_serial && vn == n && vsf == s;
					ReusedRenderFeatures = view;
					if (view) {
						check(nrf_mask_sigma_grad_src(vk, vs, n * s, 4, g_raw.data_ptr<float>(), current_stream()), "nrf_mask_sigma_grad_src");
						rc = nrf_mlp_backward_f16_lm_src(Mlp.m, vf, vcols, vs, dirs16.data_ptr(), s, g_raw.data_ptr<float>(), n * s, g_blob.data_ptr<float>(), g_x.data_ptr<float>(),
							TrainWorkspace.data_ptr(), wsb, current_stream());
					} else {
						auto feats = torch::empty({16, n * s, 2}, opt.dtype(torch::kFloat16));
						check(nrf_hash_encode_lm_f16(h, pts.data_ptr<float>(), n * s, feats.data_ptr(), keep.data_ptr<uint8_t>(), current_stream()), "nrf_hash_encode_lm_f16");
						check(nrf_mask_sigma_grad(keep.data_ptr<uint8_t>(), n * s, 4, g_raw.data_ptr<float>(), current_stream()), "nrf_mask_sigma_grad");
						rc = nrf_mlp_backward_f16_lm(Mlp.m, feats.data_ptr(), dirs16.data_ptr(), s, g_raw.data_ptr<float>(), n * s, g_blob.data_ptr<float>(), g_x.data_ptr<float>(),
							TrainWorkspace.data_ptr(), wsb, current_stream());
					}
				} else {
					torch::Tensor x = fp32_input();
					rc = nrf_mlp_backward_f16(Mlp.m, x.data_ptr<float>(), g_raw.data_ptr<float>(), n * s, g_blob.data_ptr<float>(), g_x.data_ptr<float>(), TrainWorkspace.data_ptr(), wsb,
						current_stream());
				}
				if (rc == NRF_OK) {
					uint32_t fl[2] = {0, 0};
					check(nrf_mlp_backward_f16_flags(TrainWorkspace.data_ptr(), fl, current_stream()), "nrf_mlp_backward_f16_flags");
					f16_done = !(fl[0] || fl[1]);
					if (!f16_done) { F16BackwardOverflows++; g_blob.zero_(); }
				} else if (rc != NRF_ERR_UNSUPPORTED) check(rc, "nrf_mlp_backward_f16");          // a network outside the fused kernel's family: the fp32 layer kernels
			}
			if (!f16_done) {
				PhaseScope pf("bwd.f32_chain");
				torch::Tensor x = fp32_input();
				const size_t wsb = nrf_mlp_backward_workspace_bytes(Mlp.m, n * s);
				check(nrf_mlp_backward(Mlp.m, x.data_ptr<float>(), g_raw.data_ptr<float>(), n * s, g_blob.data_ptr<float>(), g_x.data_ptr<float>(), workspace(wsb, rays.device()), wsb,
					current_stream()), "nrf_mlp_backward");
			}
			PhaseScope ph("bwd.hash_scatter");
			if (f16_done && this->EmbedFn->NFeaturesPerLevel == 2 && this->EmbedFn->Log2HashmapSize <= 19) {
				// contributions merged per table range in LDS before they reach memory (equals the packed-atomic path bit for bit; resolution 2^-30 of a level's bound)
				const size_t hwb = nrf_hash_backward_binned_workspace_bytes_for(h, n, s);
				if (!HashTrainWorkspace.defined() || (size_t)HashTrainWorkspace.numel() < hwb) HashTrainWorkspace = torch::empty({(int64_t)hwb}, opt.dtype(torch::kUInt8));
				check(nrf_hash_backward_rays_binned(h, pts.data_ptr<float>(), n, s, g_x.data_ptr<float>(), g_table.data_ptr<float>(), HashTrainWorkspace.data_ptr(), hwb, current_stream()),
					"nrf_hash_backward_rays_binned");
			} else
			check(nrf_hash_backward_rays(h, pts.data_ptr<float>(), n, s, g_x.data_ptr<float>(), g_table.data_ptr<float>(), current_stream()), "nrf_hash_backward_rays");
		}
		return {g_table, g_blob};
	}

protected:
	/// NeRFRenderer.h:164-194
	torch::Tensor RunNetwork(torch::Tensor inputs, torch::Tensor view_dirs, TNeRF fn, TEmbedder embed_fn, TEmbedDirs embeddirs_fn) override
	{
		SyncIfChanged();
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
	/// NeRFRenderer.h:530-605.  A pose render is ONE library call (nrf_render_rows: ray generation, view directions taken before any c2w_staticcam / NDC
	/// substitution, NDCRays, AABB clipping, ray-batch assembly, the Chunk loop and Near / Far -- bit-identical to the LibTorch CPU path stage by stage);
	/// an explicit ray batch is packed and handed to nrf_batchify_rays.  `sh` is taken by value: the reference keeps an ArrayRef into a tensor that its
	/// NDC branch then releases (:562 vs :567), so its own Ndc renders read freed memory at :591-600.
	/// LibraryChunkLoop = false keeps the reference's own BatchifyRays (a host loop of virtual RenderRays calls + torch::cat) in the loop instead.
	bool LibraryChunkLoop = true;
	NeRFRenderResult Render(const int h, const int w, torch::Tensor k, const NeRFRenderParams &render_params,
		std::tuple<torch::Tensor, torch::Tensor, torch::Tensor> rays = {torch::Tensor(), torch::Tensor(), torch::Tensor()},
		torch::Tensor c2w = torch::Tensor(), torch::Tensor c2w_staticcam = torch::Tensor()) override
	{
		return RenderRows(h, w, k, render_params, rays, c2w, c2w_staticcam, 0, h);
	}

	/// Rows [row0, row0 + rows) of the h x w frame seen from c2w: one rank's share of a frame (multi-GPU row tiles).  Ray r of the tile is pixel
	/// (row0 + r / w, r % w); the counter-based draws of the stochastic branches are keyed by the ray's position in the WHOLE frame, so a tile equals
	/// the corresponding slice of the full render bit for bit.  Outputs are [rows, w, ...]; Near / Far are the tile's.
	/// near_far_dev (optional): receives the tile's [min near, max far] as a DEVICE tensor and Near / Far of the result stay 0 -- the call then never waits for the GPU
	/// (a rank of a sharded render issues its ~25 launches and goes straight to the collective).
	NeRFRenderResult RenderTile(const int h, const int w, torch::Tensor k, const NeRFRenderParams &render_params, torch::Tensor c2w, const int row0, const int rows,
		torch::Tensor *near_far_dev = nullptr)
	{
		TORCH_CHECK(row0 >= 0 && rows >= 0 && row0 + rows <= h, "RenderTile: rows outside the frame");
		return RenderRows(h, w, k, render_params, {torch::Tensor(), torch::Tensor(), torch::Tensor()}, c2w, torch::Tensor(), row0, rows, near_far_dev);
	}

	/// One frame over all ranks of `comm`: this rank renders its row tile, ONE all-gather (rgb, disparity, accumulation, depth packed per pixel) returns the
	/// whole frame to every rank.  Near / Far are the whole frame's (rays are cheap: every rank generates the full ray batch for that reduction).
	NeRFRenderResult RenderSharded(const int h, const int w, torch::Tensor k, const NeRFRenderParams &render_params, torch::Tensor c2w, const TileComm &comm)
	{
		const auto [row0, rows] = comm.Rows(h);
		NeRFRenderParams rp = render_params;
		rp.ReturnRaw = false; rp.ReturnWeights = false;                 // per-sample tensors stay on the rank that made them
		torch::Tensor tile_nf;
		NeRFRenderResult tile = RenderTile(h, w, k, rp, c2w, row0, rows, &tile_nf);
		auto &o = tile.Outputs;
		auto packed = torch::cat({o.RGBMap.reshape({rows, w, 3}), o.DispMap.reshape({rows, w, 1}), o.AccMap.reshape({rows, w, 1}), o.DepthMap.reshape({rows, w, 1})}, -1).unsqueeze(0);
		auto frame = comm.AllGatherFrames(packed, h).squeeze(0);        // [h, w, 6]
		using torch::indexing::Slice;
		NeRFRenderResult res;
		res.Outputs.RGBMap = frame.index({Slice(), Slice(), Slice(0, 3)}).contiguous();
		res.Outputs.DispMap = frame.index({Slice(), Slice(), 3}).contiguous();
		res.Outputs.AccMap = frame.index({Slice(), Slice(), 4}).contiguous().reshape({-1});
		res.Outputs.DepthMap = frame.index({Slice(), Slice(), 5}).contiguous();
		// Near / Far of the whole frame (NeRFRenderer.h:602-603): min / max over the ranks' tile values would need a second collective; generating the frame's
		// rays once more on every rank is cheaper than its latency (one kernel over h*w pixels)
		const auto dev = torch::Device(torch::kCUDA, c10::hip::getCurrentHIPStream().device_index());
		nrf_view v = make_view(h, w, k, render_params, c2w, torch::Tensor(), 0, h);
		auto rays_ = torch::empty({(int64_t)h * w, v.use_viewdirs ? 11 : 8}, torch::TensorOptions().dtype(torch::kFloat32).device(dev));
		auto nf = torch::empty({2}, rays_.options());
		check(nrf_view_rays(&v, rays_.data_ptr<float>(), nf.data_ptr<float>(), current_stream()), "nrf_view_rays");
		auto nfh = nf.cpu();
		res.Near = nfh[0].item<float>(); res.Far = nfh[1].item<float>();
		return res;
	}

private:
	static nrf_view make_view(const int h, const int w, torch::Tensor k, const NeRFRenderParams &rp, torch::Tensor c2w, torch::Tensor c2w_staticcam, const int row0, const int rows)
	{
		using torch::indexing::Slice;
		nrf_view v{};
		v.h = h; v.w = w; v.row0 = row0; v.rows = rows;
		auto K = host_floats(k), M = host_floats(c2w.index({Slice(torch::indexing::None, 3), Slice(torch::indexing::None, 4)}));
		TORCH_CHECK(K.size() == 9 && M.size() == 12, "Render: k must be 3x3 and c2w at least 3x4");
		for (int i = 0; i < 9; i++) v.K[i] = K[i];
		for (int i = 0; i < 12; i++) v.c2w[i] = M[i];
		if (c2w_staticcam.defined() && c2w_staticcam.numel() != 0) {
			auto S = host_floats(c2w_staticcam.index({Slice(torch::indexing::None, 3), Slice(torch::indexing::None, 4)}));
			v.has_staticcam = 1;
			for (int i = 0; i < 12; i++) v.c2w_staticcam[i] = S[i];
		}
		v.use_viewdirs = rp.UseViewdirs; v.ndc = rp.Ndc; v.chunk = rp.Chunk;
		auto bb = host_floats(rp.BoundingBox);
		TORCH_CHECK(bb.size() == 6, "Render: BoundingBox must hold [min xyz, max xyz]");
		for (int i = 0; i < 6; i++) v.bbox[i] = bb[i];
		return v;
	}

	nrf_render_params make_params(const NeRFRenderParams &rp, torch::Tensor cone_angle, int64_t ray_base) const
	{
		nrf_render_params p{rp.NSamples, rp.NImportance, rp.LinDisp, rp.WhiteBkgr, Precision, 8};
		p.perturb = rp.Perturb; p.raw_noise_std = rp.RawNoiseStd; p.precond_alpha = rp.StochasticPreconditioningAlpha;
		if (!rp.ThinRay && cone_angle.defined() && cone_angle.numel()) { p.has_cone = 1; p.cone_angle = cone_angle.cpu().template item<float>(); }
		if (rp.BoundingBox.defined() && rp.BoundingBox.numel() == 6) { auto bb = host_floats(rp.BoundingBox); p.has_bbox = 1; for (int a = 0; a < 6; a++) p.bbox[a] = bb[a]; }
		p.seed = Seed; p.ray_base = ray_base; p.overflow_policy = OverflowPolicy;
		return p;
	}

	/// torch::linspace(0, 1, steps) on the device (NeRFRenderer.h:393, Sampler.h:21), cached: a tile render must not pay two uploads per call
	torch::Tensor linspace01(int steps, torch::Device dev)
	{
		for (auto &e : LinCache) if (e.first == steps && e.second.device() == dev) return e.second;
		LinCache.emplace_back(steps, torch::linspace(0.f, 1.f, steps, torch::kFloat).to(dev));
		return LinCache.back().second;
	}
	std::vector<std::pair<int, torch::Tensor>> LinCache;

	NeRFRenderResult RenderRows(const int h, const int w, torch::Tensor k, const NeRFRenderParams &render_params,
		std::tuple<torch::Tensor, torch::Tensor, torch::Tensor> rays, torch::Tensor c2w, torch::Tensor c2w_staticcam, const int row0, const int rows,
		torch::Tensor *near_far_dev = nullptr)
	{
		const auto dev = torch::Device(torch::kCUDA, c10::hip::getCurrentHIPStream().device_index());
		const auto opt = torch::TensorOptions().dtype(torch::kFloat32).device(dev);
		const bool from_pose = c2w.defined() && c2w.numel() != 0;
		const int stride = render_params.UseViewdirs ? 11 : 8;
		const int s = render_params.NSamples, ni = render_params.NImportance, so = ni > 0 ? s + ni : s;
		torch::Tensor rays_, cone_angle, nf;
		std::vector<int64_t> sh;
		int64_t n = 0;
		nrf_view v{};
		if (from_pose) {
			v = make_view(h, w, k, render_params, c2w, c2w_staticcam, row0, rows);
			n = (int64_t)rows * w;
			sh = {rows, w, 3};
			rays_ = torch::empty({n, stride}, opt);
			nf = torch::empty({2}, opt);
			const float px = 1.0f / v.K[0], py = 1.0f / v.K[4];                     // GetRays' cone_angle (RayUtils.h:35-43)
			cone_angle = torch::tensor(((px + py) / 2.0f) * 1.1f);
		} else {
			TORCH_CHECK(!(c2w_staticcam.defined() && c2w_staticcam.numel() != 0), "Render: c2w_staticcam replaces the camera of a POSE render (NeRFRenderer.h:554-558)");
			torch::Tensor rays_o, rays_d;
			std::tie(rays_o, rays_d, cone_angle) = rays;
			rays_o = dev_f32(rays_o); rays_d = dev_f32(rays_d);
			sh = rays_d.sizes().vec();
			auto vsrc = rays_d.reshape({-1, 3}).contiguous();                        // view directions: the un-warped rays_d (:549-561)
			if (render_params.Ndc) {
				// cone rays: the reference's scale factor is |d_ndc| / |d_ndc| = 1.0 exactly (RayUtils.h:73-81, rays_d already replaced): cone_angle keeps its value
				auto oo = torch::empty_like(rays_o), od = torch::empty_like(rays_d);
				check(nrf_ndc_rays(h, w, host_floats(k)[0], 1.f, rays_o.data_ptr<float>(), rays_d.data_ptr<float>(), rays_o.numel() / 3, oo.data_ptr<float>(), od.data_ptr<float>(),
					current_stream()), "nrf_ndc_rays");
				rays_o = oo; rays_d = od;
			}
			auto o = rays_o.reshape({-1, 3}).contiguous(), d = rays_d.reshape({-1, 3}).contiguous();
			n = o.size(0);
			auto bb = host_floats(render_params.BoundingBox);
			rays_ = torch::empty({n, stride}, opt);
			if (render_params.Ndc && render_params.UseViewdirs)
				check(nrf_pack_rays_viewsrc(o.data_ptr<float>(), d.data_ptr<float>(), vsrc.data_ptr<float>(), bb.data(), n, rays_.data_ptr<float>(), current_stream()), "nrf_pack_rays_viewsrc");
			else check(nrf_pack_rays(o.data_ptr<float>(), d.data_ptr<float>(), bb.data(), n, render_params.UseViewdirs, rays_.data_ptr<float>(), current_stream()), "nrf_pack_rays");
		}
		NeRFRenderResult all_ret;
		const bool train = WantsGrad();
		if constexpr (std::is_same_v<TEmbedder, HipHashEmbedder>) this->EmbedFn->SetTraining(train);
		SyncIfChanged();
		if (train) {
			// NeRFExecutor::Train's render (NeRFExecutor.h:876): one autograd node over the library's Chunk loop; loss.backward() then reaches the embedder's and the network's
			// parameters exactly where the reference's autograd puts their gradients
			// two configurations train: hash grid + NeRFSmall (main.cpp:220-221) and the classic HipEmbedder / HipEmbedder / NeRFImpl (a legal TNeRF of the same loop)
			torch::Tensor table_for_grad;
			if constexpr (std::is_same_v<TEmbedder, HipHashEmbedder>) {
				TORCH_CHECK(HasSmall, "training a hash-grid scene through HipNeRFRenderer needs the NeRFSmall network (nrf_mlp_backward)");
				table_for_grad = this->EmbedFn->TableForGrad();
			} else {
				TORCH_CHECK(!HasSmall, "training a positional-encoding scene through HipNeRFRenderer needs the classic NeRFImpl network");
				table_for_grad = torch::empty({0}, opt);
			}
			if (from_pose) check(nrf_view_rays(&v, rays_.data_ptr<float>(), nf.data_ptr<float>(), current_stream()), "nrf_view_rays");
			auto outs = RenderFn::apply(rays_, table_for_grad, BlobForGrad(), (int64_t)reinterpret_cast<intptr_t>(this), (int64_t)reinterpret_cast<intptr_t>(&render_params),
				(!render_params.ThinRay && cone_angle.defined() && cone_angle.numel()) ? (double)cone_angle.cpu().template item<float>() : -1.0);
			auto &o = all_ret.Outputs;
			o.RGBMap = outs[0]; o.DispMap = outs[1]; o.AccMap = outs[2]; o.DepthMap = outs[3];
			if (render_params.ReturnWeights) o.Weights = outs[4];
			if (render_params.ReturnRaw) all_ret.Raw = outs[5];
		} else if (LibraryChunkLoop) {
			auto &o = all_ret.Outputs;
			o.RGBMap = torch::empty({n, 3}, opt); o.DispMap = torch::empty({n}, opt); o.AccMap = torch::empty({n}, opt); o.DepthMap = torch::empty({n}, opt);
			if (render_params.ReturnWeights) o.Weights = torch::empty({n, (int64_t)so}, opt);
			if (render_params.ReturnRaw) all_ret.Raw = torch::empty({n, (int64_t)so, 4}, opt);
			nrf_render_outputs ro{};
			ro.d_rgb = o.RGBMap.data_ptr<float>(); ro.d_disp = o.DispMap.data_ptr<float>(); ro.d_acc = o.AccMap.data_ptr<float>(); ro.d_depth = o.DepthMap.data_ptr<float>();
			ro.d_weights = render_params.ReturnWeights ? o.Weights.data_ptr<float>() : nullptr;
			ro.d_raw = render_params.ReturnRaw ? all_ret.Raw.data_ptr<float>() : nullptr;
			nrf_render_params p = make_params(render_params, cone_angle, 0);
			torch::Tensor t = linspace01(s, dev);
			torch::Tensor u; if (ni > 0) u = linspace01(ni, dev);
			if (from_pose) {
				const size_t wsb = nrf_render_rows_workspace_bytes(Renderer, &v, &p);
				check(nrf_render_rows(Renderer, &v, &p, t.data_ptr<float>(), u.defined() ? u.data_ptr<float>() : nullptr, &ro, rays_.data_ptr<float>(), nf.data_ptr<float>(),
					workspace(wsb, dev), wsb, current_stream()), "nrf_render_rows");
			} else if (n > 0) {
				const size_t wsb = nrf_batchify_rays_workspace_bytes(Renderer, n, render_params.Chunk, &p);
				check(nrf_batchify_rays(Renderer, rays_.data_ptr<float>(), stride, n, render_params.Chunk, &p, t.data_ptr<float>(), u.defined() ? u.data_ptr<float>() : nullptr, &ro,
					workspace(wsb, dev), wsb, current_stream()), "nrf_batchify_rays");
			}
		} else {
			if (from_pose) check(nrf_view_rays(&v, rays_.data_ptr<float>(), nf.data_ptr<float>(), current_stream()), "nrf_view_rays");
			RayCursor = from_pose ? (int64_t)row0 * w : 0;
			all_ret = this->BatchifyRays(rays_, render_params.ThinRay ? torch::Tensor() : cone_angle, render_params.NSamples, render_params.Chunk,
				render_params.ReturnRaw, render_params.LinDisp, render_params.Perturb, render_params.NImportance, render_params.WhiteBkgr, render_params.RawNoiseStd,
				render_params.StochasticPreconditioningAlpha, render_params.BoundingBox, render_params.ReturnWeights);
		}
		if (all_ret.Outputs.RGBMap.defined() && all_ret.Outputs.RGBMap.numel() != 0) all_ret.Outputs.RGBMap = torch::reshape(all_ret.Outputs.RGBMap, sh);
		if (sh.size() > 2) {
			if (all_ret.Outputs.DispMap.defined() && all_ret.Outputs.DispMap.numel() != 0) all_ret.Outputs.DispMap = torch::reshape(all_ret.Outputs.DispMap, {sh[0], sh[1]});
			if (all_ret.Outputs.DepthMap.defined() && all_ret.Outputs.DepthMap.numel() != 0) all_ret.Outputs.DepthMap = torch::reshape(all_ret.Outputs.DepthMap, {sh[0], sh[1]});
		}
		// NeRFRenderResult::Near / Far are host floats (the reference's two .item() calls, :602-603): one small read-back of the device-side reduction
		if (from_pose && near_far_dev) *near_far_dev = nf;               // the caller reads (or ignores) it later: no synchronisation here
		else if (from_pose) { auto nfh = nf.cpu(); all_ret.Near = nfh[0].item<float>(); all_ret.Far = nfh[1].item<float>(); }
		else if (n > 0) check(nrf_near_far_range(rays_.data_ptr<float>(), n, stride, &all_ret.Near, &all_ret.Far, current_stream()), "nrf_near_far_range");
		return all_ret;
	}

public:
	/// NeRFRenderer.h:366-459, one fused call per chunk of packed rays
	NeRFRenderResult RenderRays(torch::Tensor ray_batch, torch::Tensor cone_angle, const int n_samples, const bool return_raw = false,
		const bool lin_disp = false, const float perturb = 0.f, const int n_importance = 0, const bool white_bkgr = false,
		const float raw_noise_std = 0.f, const float stochastic_preconditioning_alpha = 0.f, torch::Tensor bounding_box = torch::Tensor(),
		const bool return_weights = true) override
	{
		SyncIfChanged();
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
		p.seed = Seed; p.ray_base = RayCursor; RayCursor += n; p.overflow_policy = OverflowPolicy;
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

#ifdef NRFPP_WITH_LERF_RENDERER
// ---------------------------------------------------------------------------------------------------------------------
// HipLeRFRenderer : LeRFRenderer (LeRFRenderer.h:56-132).  Needs the reference's LeRFRenderer.h on the include path and LeRFRenderer.cpp in the link (the
// base class's vtable and its Render / BatchifyRays live there; that unit includes RuCLIPProcessor.h, LeRFRenderer.cpp:2, for `Relevancy`, :79 -- the external
// RuCLIP module, which is why this class can only be COMPILED, not linked, where RuCLIP is absent).  The base keeps a null CuHashEmbedder: every path that would
// touch it (RunLENetwork) is overridden.  Outputs.Relevancy is filled when prompts are set (SetLeRFPrompts): nrf_lerf_relevancy, the published LERF relevancy score that
// RuCLIP's function mirrors -- parity unpinned (nerfpp_hip.h).
// ---------------------------------------------------------------------------------------------------------------------
#include "LeRFRenderer.h"

namespace nrfpp {

class HipLeRFRenderer : public LeRFRenderer {
	HipLeRFPass Pass;

	static LeRFRendererOutputs to_ref(const LeRFPassOutputs &p)
	{
		LeRFRendererOutputs o;
		o.LangEmbedding = p.LangEmbedding; o.RenderedLangEmbedding = p.RenderedLangEmbedding; o.DispMapLE = p.DispMapLE; o.AccMapLE = p.AccMapLE;
		o.WeightsLE = p.WeightsLE; o.DepthMapLE = p.DepthMapLE; o.Relevancy = p.Relevancy;
		return o;
	}
	/// the base keeps the prompts (SetLeRFPrompts is not virtual, LeRFRenderer.h:86): hand them to the pass when they changed
	torch::Tensor SeenPos, SeenNeg;
	void SyncPrompts()
	{
		if (LerfPositives.is_same(SeenPos) && LerfNegatives.is_same(SeenNeg)) return;
		Pass.SetLeRFPrompts(LerfPositives, LerfNegatives);
		SeenPos = LerfPositives; SeenNeg = LerfNegatives;
	}
protected:
	torch::Tensor RunLENetwork(torch::Tensor inputs, LeRF lerf, CuHashEmbedder lang_embed_fn) override { RunLENetworkCalls++; return Pass.RunLENetwork(inputs); }
	LeRFRendererOutputs RawToLEOutputs(torch::Tensor raw_le, torch::Tensor z_vals_le, torch::Tensor rays_d, const int lang_embed_dim = 768, const float raw_noise_std = 0.f) override
	{
		RawToLEOutputsCalls++;
		SyncPrompts();
		auto o = to_ref(Pass.RawToLEOutputs(raw_le, z_vals_le, rays_d, lang_embed_dim, raw_noise_std));
		if (LerfPositives.defined() && LerfNegatives.defined() && LerfPositives.numel() && LerfNegatives.numel())          // LeRFRenderer.cpp:79
			o.Relevancy = Relevancy(o.RenderedLangEmbedding, LerfPositives, LerfNegatives);
		return o;
	}
public:
	/// how often the inherited torch-op path (the RNG branches of LeRFRenderer::RenderRays) landed on the two overrides above
	int64_t RunLENetworkCalls = 0, RawToLEOutputsCalls = 0;
	HipLeRFRenderer(HipHashEmbedder lang_embed_fn, LeRF lerf, torch::Tensor lerf_positives = torch::Tensor(), torch::Tensor lerf_negatives = torch::Tensor(),
		int precision = NRF_PREC_F16_SPLIT) : LeRFRenderer(CuHashEmbedder(nullptr), lerf, lerf_positives, lerf_negatives), Pass(lang_embed_fn, precision) { Pass.SyncWeights(Lerf); }

	/// after a checkpoint load or an optimizer step on the LeRF head / the language hash grid
	void SyncWeights() { Pass.SyncWeights(Lerf); }

	/// LeRFRenderer.cpp:85-187.  The deterministic render path goes to the fused matrix-core passes; the RNG branches (Perturb, RawNoiseStd, cone rays,
	/// stochastic preconditioning) take the inherited torch-op path, whose RunLENetwork / RawToLEOutputs calls land on the overrides above.
	LeRFRenderResult RenderRays(torch::Tensor ray_batch, torch::Tensor cone_angle, const int n_samples, const bool return_raw = false, const bool lin_disp = false,
		const float perturb = 0.f, const int n_importance = 0, const bool white_bkgr = false, const float raw_noise_std = 0.f, const float stochastic_preconditioning_alpha = 0.f,
		torch::Tensor bounding_box = torch::Tensor(), const bool return_weights = true) override
	{
		const bool rng = perturb > 0.f || raw_noise_std > 0.f || stochastic_preconditioning_alpha > 0.f || (cone_angle.defined() && cone_angle.numel() != 0);
		if (rng || return_raw)
			return LeRFRenderer::RenderRays(ray_batch, cone_angle, n_samples, return_raw, lin_disp, perturb, n_importance, white_bkgr, raw_noise_std,
				stochastic_preconditioning_alpha, bounding_box, return_weights);
		SyncPrompts();
		LeRFRenderResult res;
		res.Outputs = to_ref(Pass.RenderRays(ray_batch, n_samples, lin_disp, n_importance, return_weights));
		return res;
	}

	/// LeRFRenderer.cpp:265-330.  A deterministic pose render is ONE library call (nrf_lerf_render_rows: rays, the Chunk loop on the library's lanes, Relevancy,
	/// Near / Far); everything else takes the inherited path, whose RenderRays calls land on the override above.
	LeRFRenderResult Render(const int h, const int w, torch::Tensor k, const NeRFRenderParams &render_params,
		std::tuple<torch::Tensor, torch::Tensor, torch::Tensor> rays = {torch::Tensor(), torch::Tensor(), torch::Tensor()}, torch::Tensor c2w = torch::Tensor(),
		torch::Tensor c2w_staticcam = torch::Tensor()) override
	{
		const NeRFRenderParams &p = render_params;
		const bool rng = p.Perturb > 0.f || p.RawNoiseStd > 0.f || p.StochasticPreconditioningAlpha > 0.f || !p.ThinRay;
		const bool pose = c2w.defined() && c2w.numel() != 0 && !(c2w_staticcam.defined() && c2w_staticcam.numel() != 0);
		const bool batch = !(c2w.defined() && c2w.numel() != 0) && std::get<0>(rays).defined() && std::get<0>(rays).numel() != 0;
		if (!rng && batch && !p.ReturnRaw && !p.Ndc && p.NImportance > 0 && p.NSamples % 32 == 0 && (p.NSamples + p.NImportance) % 32 == 0) {
			// the TRAINING render (NeRFExecutor.h:958-961: a ray batch): the library's Chunk loop; with grad mode on one autograd node whose backward reaches Lerf's
			// parameters and the language grid's table (HipLeRFPass::LeRFRenderFn), so `lang_loss.backward()` (:981) works on the drop-in as on the reference
			SyncPrompts();
			LeRFRenderResult res;
			auto sh = std::get<1>(rays).sizes().vec();
			res.Outputs = to_ref(Pass.RenderBatch(std::get<0>(rays), std::get<1>(rays), p.BoundingBox, p.NSamples, p.NImportance, p.Chunk, p.LinDisp, p.ReturnWeights, &res.Near, &res.Far));
			if (sh.size() > 2) {                                                  // LeRFRenderer.cpp:311-319
				if (res.Outputs.DispMapLE.defined()) res.Outputs.DispMapLE = res.Outputs.DispMapLE.reshape({sh[0], sh[1]});
				if (res.Outputs.DepthMapLE.defined()) res.Outputs.DepthMapLE = res.Outputs.DepthMapLE.reshape({sh[0], sh[1]});
				if (res.Outputs.RenderedLangEmbedding.defined()) res.Outputs.RenderedLangEmbedding = res.Outputs.RenderedLangEmbedding.reshape({sh[0], sh[1], -1});
				if (res.Outputs.Relevancy.defined() && res.Outputs.Relevancy.numel() != 0) res.Outputs.Relevancy = res.Outputs.Relevancy.reshape({sh[0], sh[1], 2});
			}
			return res;
		}
		if (rng || p.ReturnRaw || p.Ndc || !pose || p.NImportance <= 0 || p.NSamples % 32 || (p.NSamples + p.NImportance) % 32)
			return LeRFRenderer::Render(h, w, k, render_params, rays, c2w, c2w_staticcam);
		SyncPrompts();
		LeRFRenderResult res;
		res.Outputs = to_ref(Pass.Render(h, w, k, p.BoundingBox, p.NSamples, p.NImportance, p.Chunk, c2w, p.UseViewdirs, p.LinDisp, p.ReturnWeights, &res.Near, &res.Far));
		// LeRFRenderer.cpp:311-328: per-pixel maps reshaped to the frame
		if (res.Outputs.RenderedLangEmbedding.defined()) res.Outputs.RenderedLangEmbedding = res.Outputs.RenderedLangEmbedding.reshape({h, w, -1});
		if (res.Outputs.Relevancy.defined() && res.Outputs.Relevancy.numel() != 0) res.Outputs.Relevancy = res.Outputs.Relevancy.reshape({h, w, 2});
		if (res.Outputs.DispMapLE.defined()) res.Outputs.DispMapLE = res.Outputs.DispMapLE.reshape({h, w});
		if (res.Outputs.DepthMapLE.defined()) res.Outputs.DepthMapLE = res.Outputs.DepthMapLE.reshape({h, w});
		return res;
	}
};

}  // namespace nrfpp
#endif  // NRFPP_WITH_LERF_RENDERER

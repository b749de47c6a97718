// adapter_util.h -- TEST INFRASTRUCTURE shared by adapter_check.cpp and adapter_bench.cpp: synthetic weights (include/nrf_synth.h), the Blender-Lego-shaped camera
// (load_blender.h:43-57,161,190-192 restated) and raw fp32 file I/O.
#pragma once
#define NRFPP_WITH_REFERENCE
#include "nerfpp_torch.h"
#include "nrf_synth.h"

#include <cmath>
#include <cstdio>
#include <string>
#include <vector>

using torch::indexing::Slice;
using torch::indexing::None;
using ClassicModel = NeRF;                 // (inside a NeRFRenderer subclass the name NeRF is the base's data member)
using ClassicRendererBase = NeRFRenderer<Embedder, Embedder, ClassicModel>;

inline void fill_synth(torch::Tensor p, uint32_t seed, float amp)
{
	torch::NoGradGuard ng;
	auto flat = torch::empty({p.numel()}, torch::kFloat32);
	float *d = flat.data_ptr<float>();
	for (int64_t i = 0; i < p.numel(); i++) d[i] = nrf_synth_sym(seed, (uint32_t)i, amp);
	p.copy_(flat.view(p.sizes()));
}

inline torch::Tensor lego_K(int h, int w)
{
	float focal = 0.5f * w / std::tan(0.5f * 0.6911112f);
	float kdata[] = {focal, 0, 0.5f * w, 0, focal, 0.5f * h, 0, 0, 1};
	return torch::from_blob(kdata, {3, 3}).clone();
}

inline torch::Tensor orbit_pose(float theta_deg, float phi_deg, float radius)
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

inline torch::Tensor read_f32(const std::string &path, std::vector<int64_t> shape)
{
	int64_t n = 1; for (auto v : shape) n *= v;
	auto t = torch::empty({n}, torch::kFloat32);
	FILE *f = fopen(path.c_str(), "rb");
	if (!f || fread(t.data_ptr<float>(), 4, (size_t)n, f) != (size_t)n) throw std::runtime_error("cannot read " + path);
	fclose(f);
	return t.view(shape);
}

inline void write_f32(const std::string &path, torch::Tensor t)
{
	auto c = t.detach().to(torch::kCPU, torch::kFloat32).contiguous();
	FILE *f = fopen(path.c_str(), "wb");
	if (!f || fwrite(c.data_ptr<float>(), 4, (size_t)c.numel(), f) != (size_t)c.numel()) throw std::runtime_error("cannot write " + path);
	fclose(f);
}


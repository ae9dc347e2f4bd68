// wnv_store.h -- host-side tensor store: ingests reference state_dict entries and folds weight norm.
#pragma once
#include <cmath>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/wnv.h"

// ------------------------------------------------------------------------------------------------
// host tensors + weight-norm folding
// ------------------------------------------------------------------------------------------------
struct HostTensor {
    std::vector<int64_t> shape;
    std::vector<float> data;
    int64_t numel() const { int64_t n = 1; for (auto s : shape) n *= s; return n; }
};

static inline bool ends_with(const std::string& s, const char* suf) {
    const size_t n = strlen(suf);
    return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}

// Accepts "x.weight", or the pair "x.weight_g" + "x.weight_v" (any order, across calls): stores the
// fused tensor under "x.weight".  w = v * (g / ||v||), norm over all dims but 0 (torch weight_norm dim=0).
struct TensorStore {
    std::map<std::string, HostTensor> fused;
    std::map<std::string, HostTensor> pending;   // weight_g / weight_v halves waiting for their partner

    void put(const wnv_tensor& t) {
        HostTensor ht;
        ht.shape.assign(t.shape, t.shape + t.ndim);
        ht.data.assign(t.data, t.data + ht.numel());
        std::string name = t.name;
        if (ends_with(name, "weight_g") || ends_with(name, "weight_v")) {
            pending[name] = std::move(ht);
            const std::string base = name.substr(0, name.size() - 2);   // "...weight"
            auto ig = pending.find(base + "_g"), iv = pending.find(base + "_v");
            if (ig != pending.end() && iv != pending.end()) {
                HostTensor w;
                w.shape = iv->second.shape;
                const int64_t co = w.shape[0], inner = iv->second.numel() / co;
                w.data.resize(iv->second.data.size());
                for (int64_t o = 0; o < co; ++o) {
                    double ss = 0.0;
                    const float* v = iv->second.data.data() + o * inner;
                    for (int64_t i = 0; i < inner; ++i) ss += (double)v[i] * v[i];
                    const float scale = ig->second.data[o] / (float)std::sqrt(ss);
                    for (int64_t i = 0; i < inner; ++i) w.data[o * inner + i] = v[i] * scale;
                }
                fused[base] = std::move(w);
                pending.erase(base + "_g");
                pending.erase(base + "_v");
            }
        } else {
            fused[name] = std::move(ht);
        }
    }
    const HostTensor* get(const std::string& n) const {
        auto it = fused.find(n);
        return it == fused.end() ? nullptr : &it->second;
    }
};


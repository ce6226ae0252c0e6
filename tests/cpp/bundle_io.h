// TEST INFRASTRUCTURE: named-array bundles exchanged with tests/test_ref_signature_gpu.py (tests/bundle_io.py writes the
// same layout): "AOSB" u32 count, then per array u16 name length, name, u8 dtype (0 u8, 1 i32, 2 f32, 3 i64, 4 f64),
// u8 ndim, u64 dims, raw little-endian data.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

struct BundleArray {
    int dtype = 0;
    std::vector<uint64_t> dims;
    std::vector<uint8_t> bytes;
    size_t count() const
    {
        size_t n = 1;
        for (uint64_t d : dims) n *= (size_t)d;
        return n;
    }
    template <typename T> const T *as() const { return reinterpret_cast<const T *>(bytes.data()); }
    template <typename T> T scalar() const { return as<T>()[0]; }
};

struct Bundle {
    std::map<std::string, BundleArray> a;
    const BundleArray &operator[](const std::string &k) const
    {
        auto it = a.find(k);
        if (it == a.end()) throw std::runtime_error("bundle: no array '" + k + "'");
        return it->second;
    }
    bool has(const std::string &k) const { return a.count(k) != 0; }
    static size_t esize(int dt) { return dt == 0 ? 1 : dt == 1 ? 4 : dt == 2 ? 4 : 8; }
    static Bundle load(const std::string &path)
    {
        FILE *f = fopen(path.c_str(), "rb");
        if (!f) throw std::runtime_error("cannot open " + path);
        char magic[4];
        uint32_t n = 0;
        if (fread(magic, 1, 4, f) != 4 || memcmp(magic, "AOSB", 4) != 0 || fread(&n, 4, 1, f) != 1) throw std::runtime_error("bad bundle");
        Bundle B;
        for (uint32_t i = 0; i < n; ++i) {
            uint16_t nl = 0;
            if (fread(&nl, 2, 1, f) != 1) throw std::runtime_error("bad bundle");
            std::string name(nl, ' ');
            if (fread(&name[0], 1, nl, f) != nl) throw std::runtime_error("bad bundle");
            uint8_t dt = 0, nd = 0;
            if (fread(&dt, 1, 1, f) != 1 || fread(&nd, 1, 1, f) != 1) throw std::runtime_error("bad bundle");
            BundleArray A;
            A.dtype = dt;
            A.dims.resize(nd);
            if (nd && fread(A.dims.data(), 8, nd, f) != nd) throw std::runtime_error("bad bundle");
            A.bytes.resize(A.count() * esize(dt) + 8);
            const size_t nb = A.count() * esize(dt);
            if (nb && fread(A.bytes.data(), 1, nb, f) != nb) throw std::runtime_error("bad bundle");
            B.a[name] = std::move(A);
        }
        fclose(f);
        return B;
    }
    template <typename T> void put(const std::string &name, int dtype, const std::vector<T> &v, std::vector<uint64_t> dims = {})
    {
        BundleArray A;
        A.dtype = dtype;
        A.dims = dims.empty() ? std::vector<uint64_t>{(uint64_t)v.size()} : dims;
        A.bytes.resize(v.size() * sizeof(T) + 8);
        if (!v.empty()) memcpy(A.bytes.data(), v.data(), v.size() * sizeof(T));
        a[name] = std::move(A);
    }
    void save(const std::string &path) const
    {
        FILE *f = fopen(path.c_str(), "wb");
        if (!f) throw std::runtime_error("cannot write " + path);
        const uint32_t n = (uint32_t)a.size();
        fwrite("AOSB", 1, 4, f);
        fwrite(&n, 4, 1, f);
        for (const auto &kv : a) {
            const uint16_t nl = (uint16_t)kv.first.size();
            fwrite(&nl, 2, 1, f);
            fwrite(kv.first.data(), 1, nl, f);
            const uint8_t dt = (uint8_t)kv.second.dtype, nd = (uint8_t)kv.second.dims.size();
            fwrite(&dt, 1, 1, f);
            fwrite(&nd, 1, 1, f);
            if (nd) fwrite(kv.second.dims.data(), 8, nd, f);
            const size_t nb = kv.second.count() * esize(dt);
            if (nb) fwrite(kv.second.bytes.data(), 1, nb, f);
        }
        fclose(f);
    }
};

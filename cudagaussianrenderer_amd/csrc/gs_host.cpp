// gs_host.cpp -- host-side parts of the path: PLY ingest, quaternion packing, SH re-layout,
// default camera, procedural scene.  No GPU code; plain C++17.
//
//   gs_parse_ply                    <- parsePly            src/PlyParser.cpp:15-339
//   gs_encode_vec4                  <- encodeVec4          src/PlyParser.cpp:155-165
//   gs_realign_spherical_harmonics  <- realignSphericalHarmonics  src/Demo.cpp:160-194
//   (gs_camera_from_bounds / gs_camera_from_pose live in gs_camera.cpp, on top of the windowless CameraControls)
//   gs_generate_random_gaussians    <- generateRandomGaussians   src/Demo.cpp:119-143
#include <fcntl.h>
#include <unistd.h>

#include <algorithm>
#include <array>
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <fstream>
#include <limits>
#include <sstream>
#include <stdexcept>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include "gsplat_hip.h"

namespace
{

inline float clamp01(float x) { return std::min(std::max(x, 0.0f), 1.0f); }

// ---------------------------------------------------------------------------------------------
// PLY: the reference reads the body with one 4-byte ifstream::read per value (62 M reads for a 1 M
// splat file); here the host threads read it in blocks and transpose them, with identical results.
// ---------------------------------------------------------------------------------------------
struct Header
{
    std::vector<std::string> properties;
    int vertexCount = -1;
    std::streampos bodyOffset = 0;
};

Header parseHeader(std::ifstream& in)
{
    Header h;
    std::string line;
    bool littleEndian = false;
    int iteration = 0;
    while (std::getline(in, line))
    {
        std::stringstream ss(line);
        std::string word;
        if (!(ss >> word)) throw std::invalid_argument("Unexpected empty line.");
        if (word == "ply") continue; // not counted against the header-length limit, as in the reference
        if (word == "format")
        {
            std::string fmt;
            littleEndian = static_cast<bool>(ss >> fmt) && fmt == "binary_little_endian";
        }
        else if (word == "element")
        {
            std::string kind, n;
            if (!(ss >> kind) || kind != "vertex") throw std::invalid_argument("Unexpected element type.");
            if (!(ss >> n)) throw std::invalid_argument("Unexpected element count.");
            h.vertexCount = std::stoi(n);
        }
        else if (word == "property")
        {
            std::string type, name;
            if (!(ss >> type) || type != "float") throw std::invalid_argument("Unexpected property format, expected float.");
            if (!(ss >> name)) throw std::invalid_argument("Unexpected property, missing name.");
            if (std::find(h.properties.begin(), h.properties.end(), name) != h.properties.end())
                throw std::invalid_argument("Duplicated property \"" + name + "\".");
            h.properties.push_back(name);
        }
        else if (word == "end_header")
        {
            if (!littleEndian || h.vertexCount == -1) throw std::invalid_argument("Invalid header. Format or vertex count not found.");
            break;
        }
        if (++iteration > 256) throw std::invalid_argument("Invalid header. End not found.");
    }
    h.bodyOffset = in.tellg();
    return h;
}

int findProperty(const std::vector<std::string>& props, const std::string& name)
{
    auto it = std::find(props.begin(), props.end(), name);
    if (it == props.end()) throw std::invalid_argument("Missing property \"" + name + "\".");
    return (int) (it - props.begin());
}

float* allocFloats(size_t n) { return static_cast<float*>(std::malloc(std::max<size_t>(n, 1) * sizeof(float))); }

// splitmix64: small, seedable, identical everywhere (the reference uses std::rand through glm).
struct SplitMix64
{
    uint64_t state;
    uint64_t next()
    {
        uint64_t z = (state += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    float uniform() { return (float) (next() >> 40) * (1.0f / 16777216.0f); } // [0,1), 24 bits
    float uniform(float lo, float hi) { return lo + (hi - lo) * uniform(); }
};

} // namespace

extern "C" uint32_t gs_encode_vec4(const float v[4])
{
    // Truncating (not rounding) 8-bit pack, x in the top byte.
    uint32_t out = 0;
    for (int i = 0; i != 4; ++i) out = (out << 8) | (uint32_t) (clamp01(v[i]) * 255.0f);
    return out;
}

extern "C" size_t gs_realigned_sh_size(int groupSize, int shCount, int splatCount)
{
    const int groups = (splatCount + groupSize - 1) / groupSize;
    return (size_t) groups * groupSize * shCount;
}

extern "C" void gs_realign_spherical_harmonics(const float* src, float* dst, int groupSize, int shCount, int splatCount)
{
    // [splat][coef] -> [group][coef][lane]; the unused tail of the last group is zero-filled.
    const size_t total = gs_realigned_sh_size(groupSize, shCount, splatCount);
    std::fill(dst, dst + total, 0.0f);
    for (int splat = 0; splat != splatCount; ++splat)
    {
        const int group = splat / groupSize, lane = splat % groupSize;
        const float* s = src + (size_t) splat * shCount;
        float* d = dst + (size_t) group * groupSize * shCount + lane;
        for (int i = 0; i != shCount; ++i) d[(size_t) i * groupSize] = s[i];
    }
}

// Storage order along a 3-D Morton (Z-order) curve through the positions' bounding box, 21 bits per axis; splats with equal codes
// keep their order (std::stable_sort on the code).  storageToSplat[k] = the caller's index of the splat stored k-th.
extern "C" void gs_morton_order(const float* position4, int32_t count, int32_t* storageToSplat)
{
    if (count <= 0) return;
    float lo[3] = {position4[0], position4[1], position4[2]}, hi[3] = {position4[0], position4[1], position4[2]};
    for (int32_t i = 0; i != count; ++i)
        for (int a = 0; a != 3; ++a)
        {
            const float v = position4[4 * (size_t) i + a];
            if (v < lo[a]) lo[a] = v; // (NaN positions never move the box; their codes below come out as 0)
            if (v > hi[a]) hi[a] = v;
        }
    auto spread = [](uint64_t v) { // 21 bits -> every third bit
        v &= 0x1FFFFFull;
        v = (v | (v << 32)) & 0x1F00000000FFFFull;
        v = (v | (v << 16)) & 0x1F0000FF0000FFull;
        v = (v | (v << 8)) & 0x100F00F00F00F00Full;
        v = (v | (v << 4)) & 0x10C30C30C30C30C3ull;
        v = (v | (v << 2)) & 0x1249249249249249ull;
        return v;
    };
    std::vector<std::pair<uint64_t, int32_t>> keyed((size_t) count);
    for (int32_t i = 0; i != count; ++i)
    {
        uint64_t code = 0;
        for (int a = 0; a != 3; ++a)
        {
            const double extent = (double) hi[a] - (double) lo[a];
            double u = extent > 0.0 ? ((double) position4[4 * (size_t) i + a] - (double) lo[a]) / extent : 0.0;
            if (!(u >= 0.0)) u = 0.0;
            if (u > 1.0) u = 1.0;
            code |= spread((uint64_t) (u * 2097151.0)) << a;
        }
        keyed[(size_t) i] = {code, i};
    }
    std::stable_sort(keyed.begin(), keyed.end(), [](const auto& x, const auto& y) { return x.first < y.first; });
    for (int32_t k = 0; k != count; ++k) storageToSplat[k] = keyed[(size_t) k].second;
}

extern "C" void gs_generate_random_gaussians(int32_t count, uint64_t seed, float minScale, float maxScale, const float minPosition[3],
                                             const float maxPosition[3], float* position4, float* scaleAndRotation4, float* color4)
{
    // Distributions of Demo.cpp:119-143: position ~ U(box), rotation = angleAxis(U[0,pi], uniform axis),
    // scale ~ U[min,max]^3, colour RGBA ~ U[0,1]^4 (alpha is the opacity used at SH degree 0), position.w = 1.
    SplitMix64 rng{seed};
    for (int32_t i = 0; i != count; ++i)
    {
        float* p = position4 + 4 * (size_t) i;
        for (int k = 0; k != 3; ++k) p[k] = rng.uniform(minPosition[k], maxPosition[k]);
        p[3] = 1.0f;
        // uniform axis on the sphere: z ~ U[-1,1], phi ~ U[0,2pi)
        const float z = rng.uniform(-1.0f, 1.0f);
        const float phi = rng.uniform(0.0f, 6.28318530717958647692f);
        const float rxy = std::sqrt(std::max(0.0f, 1.0f - z * z));
        const float axis[3] = {rxy * std::cos(phi), rxy * std::sin(phi), z};
        const float angle = rng.uniform(0.0f, 3.14159265358979323846f);
        const float s = std::sin(angle * 0.5f);
        const float quat[4] = {axis[0] * s, axis[1] * s, axis[2] * s, std::cos(angle * 0.5f)};
        const float enc[4] = {(quat[0] + 1.0f) * 0.5f, (quat[1] + 1.0f) * 0.5f, (quat[2] + 1.0f) * 0.5f, (quat[3] + 1.0f) * 0.5f};
        const uint32_t packed = gs_encode_vec4(enc);
        float* sr = scaleAndRotation4 + 4 * (size_t) i;
        for (int k = 0; k != 3; ++k) sr[k] = rng.uniform(minScale, maxScale);
        std::memcpy(&sr[3], &packed, sizeof(float));
        float* c = color4 + 4 * (size_t) i;
        for (int k = 0; k != 4; ++k) c[k] = rng.uniform();
    }
}

extern "C" int gs_parse_ply(const char* filePath, GsPly* out, char* err, size_t errLen)
{
    std::memset(out, 0, sizeof(*out));
    try
    {
        std::ifstream in(filePath, std::ios_base::binary);
        Header h = parseHeader(in);
        const size_t n = (size_t) std::max(h.vertexCount, 0);
        const size_t numProps = h.properties.size();

        // Required attributes (PlyParser.cpp:182-196).
        static const char* required[14] = {"x", "y", "z", "rot_0", "rot_1", "rot_2", "rot_3", "scale_0", "scale_1", "scale_2",
                                           "f_dc_0", "f_dc_1", "f_dc_2", "opacity"};
        int off[14];
        for (int i = 0; i != 14; ++i) off[i] = findProperty(h.properties, required[i]);

        // Optional f_rest_* and the degree they imply: 9->1, 24->2, 45->3, 72->4 (PlyParser.cpp:204-243).
        std::vector<int> rest;
        for (;;)
        {
            auto it = std::find(h.properties.begin(), h.properties.end(), "f_rest_" + std::to_string(rest.size()));
            if (it == h.properties.end()) break;
            rest.push_back((int) (it - h.properties.begin()));
        }
        const int extra = (int) rest.size();
        int expected = 0, degree = 0;
        while (expected < extra)
        {
            expected += (2 * (degree + 1) + 1) * 3;
            ++degree;
        }
        if (expected != extra)
            throw std::invalid_argument("Expected degree " + std::to_string(degree) + ", " + std::to_string(expected) +
                                        " extra spherical harmonics. Found " + std::to_string(extra) + ".");
        const int shCount = extra + 3;

        out->vertexCount = h.vertexCount;
        out->sphericalHarmonicsDegree = degree;
        out->sphericalHarmonicsCount = shCount;
        out->position = allocFloats(n * 4);
        out->scaleAndRotation = allocFloats(n * 4);
        out->color = allocFloats(n * 4);
        if (degree != 0) out->sphericalHarmonics = allocFloats(n * shCount);

        // Body: [vertex][property] little-endian floats.  The reference reads it with one 4-byte ifstream::read per value
        // (PlyParser.cpp:132-140); here every host thread reads its slice of the vertices in blocks of a few thousand rows (pread, so
        // the threads share one descriptor) and converts them while they are in its cache.  A short file leaves zeros, as a block read
        // into a zeroed buffer would.  Every vertex is converted independently and min/max are exact whatever the order: the arrays are
        // the same bytes as the serial loop's (and as the reference's, tests/test_host_cpu.py).
        const int perChannel = extra / 3;
        struct Descriptor // closed on every way out of this function, exceptions included
        {
            int fd;
            ~Descriptor() { if (fd >= 0) ::close(fd); }
        } file{::open(filePath, O_RDONLY)};
        const int fd = file.fd;
        if (fd < 0) throw std::runtime_error(std::string("Cannot open '") + filePath + "' for its vertex data: " + std::strerror(errno) + ".");
        const size_t rowBytes = numProps * sizeof(float);
        const off_t bodyOffset = (off_t) h.bodyOffset;
        const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
        const size_t threads = n < 65536 ? 1 : std::min<size_t>(std::min<unsigned>(hw, 16u), n / 32768);
        // boundsMax starts at numeric_limits<float>::min() (smallest positive), a reference quirk kept (PlyParser.cpp:290).
        std::vector<std::array<float, 6>> bounds(threads, {std::numeric_limits<float>::max(), std::numeric_limits<float>::max(), std::numeric_limits<float>::max(),
                                                           std::numeric_limits<float>::min(), std::numeric_limits<float>::min(), std::numeric_limits<float>::min()});
        const auto convertSlice = [&](size_t slice) {
            const size_t begin = n * slice / threads, end = n * (slice + 1) / threads;
            constexpr size_t kBlockRows = 4096;
            std::vector<float> block(kBlockRows * numProps);
            float* bmin = bounds[slice].data();
            float* bmax = bmin + 3;
            for (size_t first = begin; first < end; first += kBlockRows)
            {
                const size_t rows = std::min(kBlockRows, end - first);
                size_t got = 0;
                const size_t want = rows * rowBytes;
                while (got < want)
                {
                    const ssize_t r = ::pread(fd, reinterpret_cast<char*>(block.data()) + got, want - got, bodyOffset + (off_t) (first * rowBytes + got));
                    if (r <= 0) break;
                    got += (size_t) r;
                }
                if (got < want) std::memset(reinterpret_cast<char*>(block.data()) + got, 0, want - got);
                for (size_t i = first; i != first + rows; ++i)
                {
                    const float* row = block.data() + (i - first) * numProps;
                    float* p = out->position + 4 * i;
                    float* s = out->scaleAndRotation + 4 * i;
                    float* c = out->color + 4 * i;

                    // sigmoid, evaluated in double and rounded once (the reference's unqualified exp() is ::exp(double))
                    const float opacity = (float) (1.0 / (1.0 + std::exp(-(double) row[off[13]])));
                    for (int k = 0; k != 3; ++k)
                    {
                        p[k] = row[off[k]];
                        bmin[k] = std::min(bmin[k], p[k]);
                        bmax[k] = std::max(bmax[k], p[k]);
                        s[k] = std::exp(row[off[7 + k]]);
                        c[k] = row[off[10 + k]] * 0.28209479177387814f + 0.5f; // SH band 0 -> rgb, unclamped
                    }
                    p[3] = opacity;
                    c[3] = opacity;

                    // rot_0 is w. Normalise, map to [0,1], truncating 8-bit pack in (x,y,z,w) order.
                    float qw = row[off[3]], qx = row[off[4]], qy = row[off[5]], qz = row[off[6]];
                    const float len = std::sqrt((qw * qw + qx * qx) + (qy * qy + qz * qz));
                    if (len <= 0.0f)
                    {
                        qw = 1.0f;
                        qx = qy = qz = 0.0f;
                    }
                    else
                    {
                        const float inv = 1.0f / len;
                        qw *= inv;
                        qx *= inv;
                        qy *= inv;
                        qz *= inv;
                    }
                    const float enc[4] = {(qx + 1.0f) * 0.5f, (qy + 1.0f) * 0.5f, (qz + 1.0f) * 0.5f, (qw + 1.0f) * 0.5f};
                    const uint32_t packed = gs_encode_vec4(enc);
                    std::memcpy(&s[3], &packed, sizeof(float));

                    if (degree != 0)
                    {
                        // f_dc (rgb) then f_rest, which the file stores channel-major [c][k] -> [k][c] (PlyParser.cpp:269-286).
                        float* sh = out->sphericalHarmonics + i * (size_t) shCount;
                        for (int k = 0; k != 3; ++k) sh[k] = row[off[10 + k]];
                        for (int j = 0; j != perChannel; ++j)
                            for (int ch = 0; ch != 3; ++ch) sh[3 + j * 3 + ch] = row[rest[ch * perChannel + j]];
                    }
                }
            }
        };
        // A slice that cannot get a thread of its own (EAGAIN under a process or thread limit) is converted by this thread; an exception
        // inside a slice (bad_alloc for its block) is carried out of its thread and rethrown after every thread has been joined, so
        // the caller gets the error text and return code of this function's contract and never std::terminate.
        std::vector<std::exception_ptr> failed(threads);
        const auto guardedSlice = [&](size_t slice) noexcept {
            try { convertSlice(slice); }
            catch (...) { failed[slice] = std::current_exception(); }
        };
        {
            struct Pool
            {
                std::vector<std::thread> threads;
                ~Pool() { for (std::thread& t : threads) if (t.joinable()) t.join(); }
            } pool;
            pool.threads.reserve(threads);
            std::vector<size_t> inlineSlices{0};
            for (size_t t = 1; t < threads; ++t)
            {
                try { pool.threads.emplace_back(guardedSlice, t); }
                catch (const std::system_error&) { inlineSlices.push_back(t); }
            }
            for (size_t slice : inlineSlices) guardedSlice(slice);
        } // joined
        for (const std::exception_ptr& e : failed)
            if (e) std::rethrow_exception(e);
        float bmin[3], bmax[3];
        for (int k = 0; k != 3; ++k)
        {
            bmin[k] = bounds[0][(size_t) k];
            bmax[k] = bounds[0][(size_t) k + 3];
            for (size_t t = 1; t < threads; ++t)
            {
                bmin[k] = std::min(bmin[k], bounds[t][(size_t) k]);
                bmax[k] = std::max(bmax[k], bounds[t][(size_t) k + 3]);
            }
        }
        for (int k = 0; k != 3; ++k)
        {
            out->boundsMin[k] = bmin[k];
            out->boundsMax[k] = bmax[k];
        }
        return h.vertexCount;
    }
    catch (const std::exception& e)
    {
        if (err && errLen) std::snprintf(err, errLen, "%s", e.what());
        gs_free_ply(out);
        return -1;
    }
}

extern "C" void gs_free_ply(GsPly* ply)
{
    std::free(ply->position);
    std::free(ply->scaleAndRotation);
    std::free(ply->color);
    std::free(ply->sphericalHarmonics);
    ply->position = ply->scaleAndRotation = ply->color = ply->sphericalHarmonics = nullptr;
}

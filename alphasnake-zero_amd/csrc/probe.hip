// probe.hip -- the shader clock the chip holds WHILE other kernels run (MI355X lowers its clock under MFMA load; the
// roofline of csrc/conv_split.hip is quoted against the 2.4 GHz peak, so bench.py reports next to it what the chip
// actually ran at).  One wavefront, launched on a stream of its own beside the measured work: it reads the shader-cycle
// counter (s_memtime) and the constant 100 MHz counter (s_memrealtime), spins on scalar instructions for the requested
// time, and reads both again.  No vector work, no memory traffic besides its two result words: the measured kernels are
// not touched (the guide's rule: no stamp executes inside the real kernel).
#include "common.h"
#include <string.h>

__global__ __launch_bounds__(64) void k_clock_probe(unsigned long long *__restrict__ out, unsigned ticks)
{
    if (threadIdx.x != 0) return;
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = r0;
    while (r1 - r0 < ticks) {
        __builtin_amdgcn_s_sleep(32);
        r1 = __builtin_amdgcn_s_memrealtime();
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    out[0] = c1 - c0;
    out[1] = r1 - r0;
}

// d_out[0] = shader cycles, d_out[1] = 100 MHz ticks elapsed while the probe sat on its compute unit for ~microseconds
extern "C" int snk_clock_probe(uint64_t *d_out, int microseconds, void *stream)
{
    SNK_REQUIRE(d_out && microseconds > 0 && microseconds <= 100000, "snk_clock_probe: bad argument");
    k_clock_probe<<<1, 64, 0, (hipStream_t)stream>>>((unsigned long long *)d_out, (unsigned)microseconds * 100u);
    SNK_CHECK_HIP(hipGetLastError());
    return 0;
}

// sha-256 (hex) of a kernel source file as it was when this library was built ("conv_split.hip", "engine.hip", "common.h", ...),
// NULL for a name the library does not know.  profiles/*.json name the hashes of the sources they were measured on.
extern "C" const char *snk_source_hash(const char *source_file)
{
    static const struct { const char *name, *sha256; } ids[] = {
#include "build_id.h"
    };
    if (!source_file) return nullptr;
    for (const auto &e : ids)
        if (!strcmp(e.name, source_file)) return e.sha256;
    return nullptr;
}

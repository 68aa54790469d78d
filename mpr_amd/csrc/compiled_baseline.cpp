/*
 * compiled_baseline.cpp — the "compiled expression" comparison point of the reference's paper
 * (benchmark/dump_tape.cpp:43-171 prints a tape as straight-line CUDA, benchmark/brute.cu:29-62
 * is such a kernel, pasted in by hand and compiled offline): here the tape is turned into
 * straight-line HIP source and compiled for the device at run time with hiprtc, then evaluated
 * for every pixel of the S x S image without any hierarchy (the same pixels and coordinates as
 * render2D_brute, src/context.cu:1461-1508).
 *
 * The generated arithmetic is, clause by clause, float_clause() of device_math.hpp with the
 * functions of include/mpr_fmath.h (the header text is embedded at build time) and is compiled
 * without FMA contraction, so the image equals render2D_brute's — which is what the test checks.
 * Slots become local variables; the compiler sees one basic block per pixel.
 */
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/mpr_amd.h"
#include "../../include/mpr_clause.h"
#include "internal.hpp"

static const char* const k_fmath_src =
#include "build/mpr_fmath_src.inc"
    ;

namespace {

std::string fhex(uint32_t bits)
{
    char b[48];
    snprintf(b, sizeof b, "__uint_as_float(0x%08xu)", bits);
    return b;
}

/* straight-line source of one pixel's evaluation: one float variable per slot */
std::string generate(const std::vector<uint64_t>& cl, int nslots)
{
    std::string s;
    s += "#include \"mpr_fmath.h\"\n";
    s += "extern \"C\" __global__ void mpr_compiled(int S, const float* __restrict__ mat, float zc, int* __restrict__ image)\n{\n";
    s += "    const int px = threadIdx.x + blockIdx.x * blockDim.x, py = threadIdx.y + blockIdx.y * blockDim.y;\n";
    s += "    if (px >= S || py >= S) return;\n";
    s += "    const float size_recip = 1.0f / (float)(unsigned)S;\n";
    s += "    const float fx = ((px + 0.5f) * size_recip - 0.5f) * 2.0f;\n";
    s += "    const float fy = ((py + 0.5f) * size_recip - 0.5f) * 2.0f;\n";
    s += "    const float fw = mat[2] * fx + mat[5] * fy + mat[8];\n";
    s += "    const float vx = (mat[0] * fx + mat[3] * fy + mat[6]) / fw;\n";
    s += "    const float vy = (mat[1] * fx + mat[4] * fy + mat[7]) / fw;\n";
    s += "    const float vz = zc;\n";
    for (int k = 0; k < nslots; ++k) s += "    float s" + std::to_string(k) + " = 0.0f;\n";
    const uint64_t head = cl.front();
    s += "    s" + std::to_string(mpr_cl_out(head)) + " = vx;\n";
    s += "    s" + std::to_string(mpr_cl_lhs(head)) + " = vy;\n";
    s += "    s" + std::to_string(mpr_cl_rhs(head)) + " = vz;\n";
    for (size_t i = 1; i + 1 < cl.size(); ++i) {
        const uint64_t c = cl[i];
        const std::string o = "s" + std::to_string(mpr_cl_out(c)), l = "s" + std::to_string(mpr_cl_lhs(c)),
                          r = "s" + std::to_string(mpr_cl_rhs(c)), imm = fhex(mpr_cl_immbits(c));
        std::string e;
        switch (mpr_cl_op(c)) {
            case MPR_OP_SQUARE_LHS: e = l + " * " + l; break;
            case MPR_OP_SQRT_LHS: e = "__builtin_sqrtf(" + l + ")"; break;
            case MPR_OP_NEG_LHS: e = "-" + l; break;
            case MPR_OP_SIN_LHS: e = "mpr_sinf(" + l + ")"; break;
            case MPR_OP_COS_LHS: e = "mpr_cosf(" + l + ")"; break;
            case MPR_OP_ASIN_LHS: e = "mpr_asinf(" + l + ")"; break;
            case MPR_OP_ACOS_LHS: e = "mpr_acosf(" + l + ")"; break;
            case MPR_OP_ATAN_LHS: e = "mpr_atanf(" + l + ")"; break;
            case MPR_OP_EXP_LHS: e = "mpr_expf(" + l + ")"; break;
            case MPR_OP_ABS_LHS: e = "__builtin_fabsf(" + l + ")"; break;
            case MPR_OP_LOG_LHS: e = "mpr_logf(" + l + ")"; break;
            case MPR_OP_ADD_LHS_IMM: e = l + " + " + imm; break;
            case MPR_OP_ADD_LHS_RHS: e = l + " + " + r; break;
            case MPR_OP_MUL_LHS_IMM: e = l + " * " + imm; break;
            case MPR_OP_MUL_LHS_RHS: e = l + " * " + r; break;
            case MPR_OP_MIN_LHS_IMM: e = "mpr_fminf(" + l + ", " + imm + ")"; break;
            case MPR_OP_MIN_LHS_RHS: e = "mpr_fminf(" + l + ", " + r + ")"; break;
            case MPR_OP_MAX_LHS_IMM: e = "mpr_fmaxf(" + l + ", " + imm + ")"; break;
            case MPR_OP_MAX_LHS_RHS: e = "mpr_fmaxf(" + l + ", " + r + ")"; break;
            case MPR_OP_SUB_LHS_IMM: e = l + " - " + imm; break;
            case MPR_OP_SUB_IMM_RHS: e = imm + " - " + r; break;
            case MPR_OP_SUB_LHS_RHS: e = l + " - " + r; break;
            case MPR_OP_DIV_LHS_IMM: e = l + " / " + imm; break;
            case MPR_OP_DIV_IMM_RHS: e = imm + " / " + r; break;
            case MPR_OP_DIV_LHS_RHS: e = l + " / " + r; break;
            case MPR_OP_COPY_IMM: e = imm; break;
            case MPR_OP_COPY_LHS: e = l; break;
            case MPR_OP_COPY_RHS: e = r; break;
            default: e = "__uint_as_float(0x7fc00000u)"; break;
        }
        s += "    " + o + " = " + e + ";\n";
    }
    s += "    if (s" + std::to_string(mpr_cl_out(cl.back())) + " < 0.0f) image[px + py * S] = 1;\n}\n";
    return s;
}

}  // namespace

struct mpr_compiled {
    int device = 0;
    hipModule_t module = nullptr;
    hipFunction_t fn = nullptr;
    float* mat_dev = nullptr;
    std::string source, log;
};

extern "C" {

int mpr_compiled_create(int32_t device, const mpr_tape* tape, mpr_compiled** out)
{
    if (!tape || !out) return mpr::set_error(MPR_ERR_INVALID, "null argument");
    for (size_t i = 1; i + 1 < tape->clauses.size(); ++i)
        if (mpr_cl_op(tape->clauses[i]) < 2) return mpr::set_error(MPR_ERR_UNSUPPORTED, "tape with jumps cannot be compiled");
    if (hipSetDevice(device) != hipSuccess) return mpr::set_error(MPR_ERR_NO_DEVICE, "hipSetDevice failed");
    mpr_compiled* k = new mpr_compiled();
    k->device = device;
    k->source = generate(tape->clauses, tape->num_slots);
    hiprtcProgram prog;
    /* hiprtc has no C library headers; what mpr_fmath.h takes from them exists as device builtins */
    const char* hdr_src[4] = {k_fmath_src,
                              "#pragma once\ntypedef unsigned int uint32_t;\ntypedef int int32_t;\n"
                              "typedef unsigned long long uint64_t;\ntypedef long long int64_t;\n",
                              "#pragma once\n#define memcpy __builtin_memcpy\n", "#pragma once\n"};
    const char* hdr_name[4] = {"mpr_fmath.h", "stdint.h", "string.h", "math.h"};
    if (hiprtcCreateProgram(&prog, k->source.c_str(), "mpr_compiled.hip", 4, hdr_src, hdr_name) != HIPRTC_SUCCESS) {
        delete k;
        return mpr::set_error(MPR_ERR_UNSUPPORTED, "hiprtcCreateProgram failed");
    }
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, device);
    const std::string arch = std::string("--offload-arch=") + prop.gcnArchName;
    const char* opts[] = {arch.c_str(), "-O3", "-ffp-contract=off", "-std=c++17"};
    const hiprtcResult cr = hiprtcCompileProgram(prog, 4, opts);
    size_t ls = 0;
    (void)hiprtcGetProgramLogSize(prog, &ls);
    if (ls > 1) {
        k->log.resize(ls);
        (void)hiprtcGetProgramLog(prog, &k->log[0]);
    }
    if (cr != HIPRTC_SUCCESS) {
        const std::string msg = "hiprtc compilation failed: " + k->log.substr(0, 2000);
        (void)hiprtcDestroyProgram(&prog);
        delete k;
        return mpr::set_error(MPR_ERR_UNSUPPORTED, msg);
    }
    size_t cs = 0;
    (void)hiprtcGetCodeSize(prog, &cs);
    std::vector<char> code(cs);
    (void)hiprtcGetCode(prog, code.data());
    (void)hiprtcDestroyProgram(&prog);
    if (hipModuleLoadData(&k->module, code.data()) != hipSuccess ||
        hipModuleGetFunction(&k->fn, k->module, "mpr_compiled") != hipSuccess ||
        hipMalloc((void**)&k->mat_dev, 9 * sizeof(float)) != hipSuccess) {
        if (k->module) (void)hipModuleUnload(k->module);
        delete k;
        return mpr::set_error(MPR_ERR_NO_DEVICE, "loading the compiled kernel failed");
    }
    *out = k;
    return MPR_OK;
}

void mpr_compiled_destroy(mpr_compiled* k)
{
    if (!k) return;
    (void)hipSetDevice(k->device);
    if (k->mat_dev) (void)hipFree(k->mat_dev);
    if (k->module) (void)hipModuleUnload(k->module);
    delete k;
}

const char* mpr_compiled_source(const mpr_compiled* k) { return k ? k->source.c_str() : nullptr; }

/* evaluates every pixel into dev_image (S*S int32, zeroed first): 1 inside, 0 outside; blocking */
int mpr_compiled_render2d(mpr_compiled* k, int32_t size, const float mat3_colmajor[9], float z, int32_t* dev_image)
{
    if (!k || !mat3_colmajor || !dev_image || size <= 0) return mpr::set_error(MPR_ERR_INVALID, "bad argument");
    if (hipSetDevice(k->device) != hipSuccess) return mpr::set_error(MPR_ERR_NO_DEVICE, "hipSetDevice failed");
    if (hipMemcpy(k->mat_dev, mat3_colmajor, 9 * sizeof(float), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemset(dev_image, 0, (size_t)size * size * sizeof(int32_t)) != hipSuccess)
        return mpr::set_error(MPR_ERR_NO_DEVICE, "hip memory operation failed");
    int S = size;
    float zc = z;
    void* args[] = {&S, &k->mat_dev, &zc, &dev_image};
    const unsigned g = (unsigned)(size + 15) / 16;
    if (hipModuleLaunchKernel(k->fn, g, g, 1, 16, 16, 1, 0, nullptr, args, nullptr) != hipSuccess ||
        hipDeviceSynchronize() != hipSuccess)
        return mpr::set_error(MPR_ERR_NO_DEVICE, "launch of the compiled kernel failed");
    return MPR_OK;
}

}  // extern "C"

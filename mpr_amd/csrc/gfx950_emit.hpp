/* gfx950_emit.hpp — instruction words of the GFX9 family (gfx950) as the host-side code generators write them (tile_gen.cpp: the
 * interval walks and the Deriv walk; voxel_gen.cpp: the float pass).  tests/test_tile_gen.py and tests/test_voxel_gen.py
 * disassemble what is emitted with the ROCm assembler and compare it with the instructions it is meant to be. */
#pragma once
#include <cstdint>
#include <vector>

namespace mpr {
namespace gfx {

constexpr uint32_t SIGN = 0x80000000u;
/* VOP2 opcodes */
constexpr int V_CNDMASK = 0, V_ADD_F32 = 1, V_SUB_F32 = 2, V_SUBREV_F32 = 3, V_MUL_F32 = 5, V_LSHLREV = 18, V_AND = 19, V_OR = 20,
              V_XOR = 21, V_ADD_U32 = 52, V_SUB_U32 = 53;
/* VOP3 opcodes */
constexpr int V3_CNDMASK = 0x100, V3_ADD_F32 = 0x101, V3_MUL_F32 = 0x105, V3_BFE_U32 = 0x1C8, V3_LSHL_OR = 0x200;
/* VOPC opcodes */
constexpr int VC_EQ_U32 = 0xCA, VC_NE_U32 = 0xCD, VC_LT_F32 = 0x41, VC_GE_F32 = 0x46;
constexpr uint32_t VCC = 106, EXEC = 126, LIT = 255;

struct Emit {
    std::vector<uint32_t>& c;
    static uint32_t V(int r) { return 256u + (uint32_t)r; }
    static uint32_t I(int k) { return 128u + (uint32_t)k; }                     /* inline integer 0..64 */
    void d(uint32_t x) { c.push_back(x); }
    void vop2(int op, int vdst, uint32_t src0, int vsrc1) { d((uint32_t)op << 25 | (uint32_t)vdst << 17 | (uint32_t)vsrc1 << 9 | src0); }
    void vop2_lit(int op, int vdst, uint32_t lit, int vsrc1) { vop2(op, vdst, LIT, vsrc1); d(lit); }
    void mov(int vdst, uint32_t src0) { d(0x7E000200u | (uint32_t)vdst << 17 | src0); }
    void mov_lit(int vdst, uint32_t lit) { mov(vdst, LIT); d(lit); }
    void vop3(int op, uint32_t dst, uint32_t s0, uint32_t s1, uint32_t s2, uint32_t neg = 0)
    {
        d(0xD0000000u | (uint32_t)op << 16 | dst);
        d(s0 | s1 << 9 | s2 << 18 | neg << 29);
    }
    void vopc(int op, uint32_t src0, int vsrc1) { d(0x7C000000u | (uint32_t)op << 17 | (uint32_t)vsrc1 << 9 | src0); }
    void swappc(int ret, int target) { d(0xBE801E00u | (uint32_t)ret << 16 | (uint32_t)target); }
    void setpc(int target) { d(0xBE801D00u | (uint32_t)target); }
    void mov_exec(uint32_t src) { d(0xBE800100u | EXEC << 16 | src); }
    void cbranch_vccz(int dwords) { d(0xBF860000u | (uint32_t)(dwords & 0xFFFF)); }
    void nop(int n) { d(0xBF800000u | (uint32_t)n); }
    void store_x2(int vaddr, int vdata, int saddr) { d(0xDC748000u); d((uint32_t)vaddr | (uint32_t)vdata << 8 | (uint32_t)saddr << 16); }
    /* quad_perm:[3,3,3,3] on src0 (the value component of a Deriv to its quad).  A register a VALU instruction wrote needs two
     * wait states before a DPP operand reads it: `w1` / `w2` = what the last / the one before last instruction wrote */
    int w1 = -1, w2 = -1;
    void wrote(int v) { w2 = w1; w1 = v; }
    void dpp_wait(int src)
    {
        if (src == w1) { nop(1); w1 = w2 = -1; }
        else if (src == w2) { nop(0); w1 = w2 = -1; }
    }
    void vop2_q3(int op, int vdst, int src0, int vsrc1)
    {
        dpp_wait(src0);
        vop2(op, vdst, 250, vsrc1);
        d(0xFF00FF00u | (uint32_t)src0);
        wrote(vdst);
    }
    void mov_q3(int vdst, int src0)
    {
        dpp_wait(src0);
        mov(vdst, 250);
        d(0xFF00FF00u | (uint32_t)src0);
        wrote(vdst);
    }
    void sop2_vcc(int op, int ssrc1) { d(0x80000000u | (uint32_t)op << 23 | VCC << 16 | (uint32_t)ssrc1 << 8 | VCC); }   /* vcc = vcc op s[ssrc1:+1] */
};

}  // namespace gfx
}  // namespace mpr

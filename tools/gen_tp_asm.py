#!/usr/bin/env python3
"""Generates apla_amd/csrc/gemm_tp_asm.inc: the compute period of the tile-alternating GEMM (gemm_tp.hip) as ONE inline-asm block
per wave position LW (0..3).

Why assembly.  The period's loop holds 160 accumulators, 52 fragment registers and the LDS-DMA offsets, and every formulation of it
in HIP source ended with hipcc renaming accumulators between K-steps, parking loop-carried fragments in the accumulator registers it
believed free, or spilling (gemm_tp.hip has the list).  Written out, the loop is short and regular:

    per K-step:  s_barrier
                 7 x [ 5 MFMA (W fragment j x the five A fragments); LDS-DMA piece j behind product LW; ds_read W fragment j of the next K-step ]
                 1 x [ 5 x ( MFMA (W fragment 7 x A fragment i); ds_read A fragment i of the next K-step ) ; ds_read W fragment 7 ]
                 operand pointers += one K-step (or: switch to the next tile's); s_waitcnt vmcnt(14)

Register map (literal names; the C++ side clobbers v0-v53 and treats a[0:159] as this block's and tp_acc_take's):
    a[4 (8 i + j) .. +3]  accumulator (i, j)          v[4 j .. +3]  W fragment j        v[32 + 4 i .. +3]  A fragment i
    v52 / v53             LDS addresses of the NEXT K-step's W / A fragments
    v54-v57 / v58-v60     lane offsets of this wave's four W / three A LDS-DMA pieces (advance by one K-step per stage)
    s[86:90] scratch, s91 ring slot, s92 K-step of the next stage, s[94:95] / s[96:97] W / A tile base

    python3 tools/gen_tp_asm.py > apla_amd/csrc/gemm_tp_asm.inc
"""
R = 5                 # ring slots
TSTG = 26624          # bytes per stage
TA_BYTES = 10240


def acc(i, j):
    b = 4 * (8 * i + j)
    return f"a[{b}:{b + 3}]"


def wf(j):
    return f"v[{4 * j}:{4 * j + 3}]"


def af(i):
    return f"v[{32 + 4 * i}:{32 + 4 * i + 3}]"


def step(lines, lw, issue, tag):
    A = lines.append
    A("s_barrier")
    # ring bookkeeping: s86 = slot of K-step s+1 (fragments to read), s87 = slot of K-step s-1 (LDS-DMA target), ring <- s86
    A("s_add_i32 s86, s91, 1")
    A(f"s_cmp_eq_u32 s86, {R}")
    A("s_cselect_b32 s86, 0, s86")
    if issue:
        A("s_add_i32 s87, s91, -1")
        A("s_cmp_lt_i32 s87, 0")
        A(f"s_cselect_b32 s87, {R - 1}, s87")
        A(f"s_mul_i32 s87, s87, {TSTG}")
        A("s_add_i32 s88, s87, %[ldsw]")     # LDS address of this wave's first W piece in the target slot
        A("s_add_i32 s89, s87, %[ldsa]")     # ... first A piece
    A("s_mov_b32 s91, s86")
    A(f"s_mul_i32 s86, s86, {TSTG}")
    A("v_add_u32_e32 v52, s86, %[woff]")
    A("v_add_u32_e32 v53, s86, %[aoff]")
    # fragments of this K-step were read during the previous one (or by the prologue): in LDS order W0..W6, A0..A4, W7.
    # Group 0 needs W0 and A0..A4: everything but the last read.
    A("s_waitcnt lgkmcnt(1)")
    for j in range(7):
        if j == 6:
            # W7 of this K-step: the last read of the previous K-step, now followed by this K-step's reads of W0..W5 (6 newer reads)
            pass
        for i in range(5):
            A(f"\" TP_MFMA_OP \" {acc(i, j)}, {wf(j)}, {af(i)}, {acc(i, j)}")
            if issue and i == lw:
                if j < 4:
                    A(f"s_add_i32 m0, s88, {j * 1024}")
                    A("s_nop 0")
                    A(f"global_load_lds_dwordx4 v{54 + j}, s[94:95]")
                else:
                    imm = (j - 4) * 1024
                    if lw >= 2 and j == 6:
                        imm -= 1024          # waves 2 and 3 own two A pieces: the third issue repeats the second
                    A(f"s_add_i32 m0, s89, {imm}")
                    A("s_nop 0")
                    A(f"global_load_lds_dwordx4 v{58 + j - 4}, s[96:97]")
        A(f"ds_read_b128 {wf(j)}, v52 offset:{j * 1024}")
    # group 7 needs W7 (previous K-step's last read): 7 newer reads (W0..W6 of the next K-step) may stay in flight
    A("s_waitcnt lgkmcnt(7)")
    for i in range(5):
        A(f"\" TP_MFMA_OP \" {acc(i, 7)}, {wf(7)}, {af(i)}, {acc(i, 7)}")
        A(f"ds_read_b128 {af(i)}, v53 offset:{i * 1024}")
    A(f"ds_read_b128 {wf(7)}, v52 offset:{7 * 1024}")
    if issue:
        # the next stage: one K-step on, or K-step 0 of the next tile (pointers, clamped A offsets and bias piece prepared by the C++ side)
        A("s_add_i32 s92, s92, 1")
        A("s_cmp_eq_u32 s92, %[nk]")
        A(f"s_cbranch_scc1 .Ltp_switch_{tag}_%=")
        for j in range(4):
            A(f"v_add_u32_e32 v{54 + j}, %[wkstep], v{54 + j}")      # the K offset lives in the lane offsets (32-bit), the tile bases stay put
        for j in range(3):
            A(f"v_add_u32_e32 v{58 + j}, %[akstep], v{58 + j}")
        A(f"s_branch .Ltp_adv_{tag}_%=")
        A(f".Ltp_switch_{tag}_%=:")
        A("s_mov_b32 s92, 0")
        A("s_mov_b64 s[94:95], %[wpn]")
        A("s_mov_b64 s[96:97], %[apn]")
        for j in range(4):
            A(f"v_subrev_u32_e32 v{54 + j}, %[wback], v{54 + j}")     # back to K-step 0: minus (nk - 1) K-steps
        A("v_mov_b32_e32 v58, %[avn0]")
        A("v_mov_b32_e32 v59, %[avn1]")
        A("v_mov_b32_e32 v60, %[avn2]")
        A("s_cmp_eq_u64 %[bpn], 0")
        A(f"s_cbranch_scc1 .Ltp_adv_{tag}_%=")
        A("s_mov_b32 m0, %[blds]")
        A("s_nop 0")
        A("global_load_lds_dwordx4 %[blane], %[bpn]")
        A(f".Ltp_adv_{tag}_%=:")
        A("s_waitcnt vmcnt(14)")
    else:
        A("s_waitcnt vmcnt(0)")


def block(lw):
    L = []
    A = L.append
    A("s_setprio 1")
    # working copies of the operands the block changes (all operands are inputs: an asm statement takes 30 operands, "+" ones count twice)
    A("s_mov_b32 s91, %[ring]")
    A("s_mov_b32 s92, %[dk]")
    A("s_mov_b64 s[94:95], %[wp]")
    A("s_mov_b64 s[96:97], %[ap]")
    for j in range(4):
        A(f"v_mov_b32_e32 v{54 + j}, %[wv{j}]")
    for j in range(3):
        A(f"v_mov_b32_e32 v{58 + j}, %[av{j}]")
    # the first K-step's fragments (its stage has landed: it was published by the barrier of the previous K-step)
    A(f"s_mul_i32 s86, s91, {TSTG}")
    A("v_add_u32_e32 v52, s86, %[woff]")
    A("v_add_u32_e32 v53, s86, %[aoff]")
    for j in range(7):
        A(f"ds_read_b128 {wf(j)}, v52 offset:{j * 1024}")
    for i in range(5):
        A(f"ds_read_b128 {af(i)}, v53 offset:{i * 1024}")
    A(f"ds_read_b128 {wf(7)}, v52 offset:{7 * 1024}")
    # loop 1: K-steps that issue a stage
    A("s_mov_b32 s90, %[n1]")
    A("s_cmp_eq_u32 s90, 0")
    A("s_cbranch_scc1 .Ltp_l1_done_%=")
    A(".Ltp_l1_%=:")
    step(L, lw, True, "a")
    A("s_add_i32 s90, s90, -1")
    A("s_cmp_lg_u32 s90, 0")
    A("s_cbranch_scc1 .Ltp_l1_%=")
    A(".Ltp_l1_done_%=:")
    # loop 2: the last K-steps of the workgroup's last tiles (nothing left to issue)
    A("s_mov_b32 s90, %[n2]")
    A("s_cmp_eq_u32 s90, 0")
    A("s_cbranch_scc1 .Ltp_l2_done_%=")
    A(".Ltp_l2_%=:")
    step(L, lw, False, "b")
    A("s_add_i32 s90, s90, -1")
    A("s_cmp_lg_u32 s90, 0")
    A("s_cbranch_scc1 .Ltp_l2_%=")
    A(".Ltp_l2_done_%=:")
    A("s_waitcnt lgkmcnt(0)")      # (the reads of a K-step that does not follow: not used)
    A("s_nop 15")                  # the last products' results before any reader
    A("s_setprio 0")
    return L


def main():
    print("// GENERATED by tools/gen_tp_asm.py — do not edit.  The compute period of gemm_tp.hip, one asm template per wave position LW.")
    print("// Operands: see compute_period in gemm_tp.hip; register map and the why: tools/gen_tp_asm.py.")
    for lw in range(4):
        print(f"#define TP_COMPUTE_ASM_LW{lw} \\")
        lines = block(lw)
        for k, ln in enumerate(lines):
            end = " \\" if k + 1 < len(lines) else ""
            print(f'  "{ln}\\n\\t"{end}')
        print()


if __name__ == "__main__":
    main()

// amx_dev.h — device-side model image / launch plan shared by amx_api.cpp (host) and amx_kernels.hip.
// Internal to ambersim_amd/csrc; the public C ABI is include/amx.h.
#pragma once
#include <stdint.h>

#define AMX_WAVE 64
#ifndef AMX_WAVES_PER_BLOCK
#define AMX_WAVES_PER_BLOCK 8  // largest workgroup in wavefronts; they share one LDS copy of the model constants
#endif
#ifndef AMX_MIN_WAVES_PER_SIMD
#define AMX_MIN_WAVES_PER_SIMD 2  // launch bound: 256 VGPRs.  4096 envs give at most 2 waves/SIMD at G=32 and LDS allows no more at G=64, and 168/128 spill
#endif
// Rotation of row i of a matrix stored with rotated 16-byte chunks (mrot; nch chunks per row): chunk c of row i sits at position
// (c + AMX_ROW_ROT(i, nch)) mod nch.  A ds_read_b128 of a wave is served in four groups of sixteen lanes that are NOT
// consecutive - {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32 - and the sixteen rows of a group must fall into
// sixteen different 4-bank slots.  With 32-float rows (nch = 8) the slot of lane i is 8 (i & 1) + (c + rot) mod 8: the rotation
// (i >> 1) mod 8 makes the eight even and the eight odd lanes of every such group distinct; the plain i mod 8 of round 2 was
// designed for eight CONSECUTIVE lanes and collided two ways in every group (r03 PMC: LDS bank-conflict cycles 12.8 % -> see
// profiles/r03_rocprofv3_pmc_biped.csv).
#define AMX_ROW_ROT(i, nch) ((((nch) == 8) ? ((i) >> 1) : (i)) % (nch))
#define AMX_MAX_NV 32           // register-row solve capacity (one matrix row per lane, <=32 columns; one 32x32 MFMA block)

// Record strides (floats) of the constant image.  All odd, so lane-per-record LDS reads are conflict free.
// Record strides (floats) of the per-environment slab, also odd.
enum { BODY_E = 29, JNT_E = 7, DOF_E = 17, ROW_E = 3, CON_E = 5, GEO_E = 13 };
enum { BODY_S = 25, JNT_S = 13, DOF_S = 13, ACT_S = 13, GEOM_S = 13, PAIR_S = 33, LIM_S = 15, EQ_S = 21, HULL_S = 7, FRI_S = 5 };
// convex-convex limits and scratch: world vertices of both hulls [2][cx_maxv][3] (cx_maxv = the model's largest hull in such a pair), two clip
// polygons [2][CX_MAXP][3], one spare word
enum { CX_MAXV = 64, CX_MAXF = 128, CX_MAXE = 192, CX_MAXP = 16, CX_TAIL = 6 * CX_MAXP + 1 };
// BodyRec: 0 parent 1 level 2 rootslot 3 jntadr 4 jntnum 5 childadr 6 childnum 7-9 pos 10-13 quat 14-16 ipos 17-20 iquat 21 mass 22-24 inertia
// JntRec : 0 type 1 qadr 2 dadr 3 body 4-6 pos 7-9 axis 10 qpos0 11 stiffness 12 springref
// DofRec : 0 body 1 jnt 2 kind(0 free-trans,1 free/ball-rot,2 hinge,3 slide) 3 axis 4 armature 5 damping 6 actadr 7 actnum 8 rootslot
//          9 first dof of the joint 10 lane of the limit row on this dof (-1: none) 11 lane of its friction-loss row (-1: none) 12 pad
// FriRec : 0 dadr 1 D = 1 / max(invweight0 (1 - imp) / imp, MINVAL) with imp = impedance at pos 0 2 b 3 frictionloss 4 pad   (all folded on the host)
// ActRec : 0 qadr 1 dadr 2 flags(1 ctrllimited,2 forcelimited,4 affine bias) 3 gear 4 gain 5-7 bias 8-9 ctrlrange 10-11 forcerange 12 pad
// GeomRec: 0 body 1 type 2-4 pos 5-8 quat 9-11 size 12 pad
// PairRec: 0 g1 1 g2 2 fn 3 conadr 4 ncon 5 vertadr 6 vertnum 7 root1 8 root2 9 dofadr 10 dofnum 11 mu 12 k 13 b 14 dmin 15 dmax
//          16 1/width 17 mid 18 power 19 ca 20 cb 21 margin 22 invw 23 hadr 24 hnum 25 hull record of geom1 26 of geom2 (fn 6) 27-32 pad
//          fn 8 / 9 (rows of a connect / weld equality, carried as pseudo-pairs): 0 body1 1 body2 5 weld? 11 torquescale 22 invweight0 sum
//          (8: translational, 9: rotational) 23-25 point in body1's frame 26-28 point in body2's frame 29-32 relpose quaternion
// HullRec: 0 nvert 1 vertices [nvert][3] 2 nface 3 face normals [nface][3] 4 face polygons [nface][9] (count, 8 vertex ids) 5 nedge 6 edges [nedge][2]
// LimRec : 0 qadr 1 dadr 2 range0 3 range1 4 margin 5 k 6 b 7 dmin 8 dmax 9 1/width 10 mid 11 power 12 ca 13 cb 14 invw
// EqRec  : 0 q1 1 d1 2 q2 3 d2 4-8 polycoef 9 k 10 b 11 dmin 12 dmax 13 1/width 14 mid 15 power 16 ca 17 cb 18 invw 19 qref1 20 qref2

// Everything the kernel needs besides the constant image: sizes, options, and a few base offsets.
// The integer fields are listed ONCE, here: DevModel, the model-view accessors (amx_kernels.hip: MV<S>), the plan dump that
// tools/gen_specs.py turns into amx_specs_gen.h and the specialisation match in amx_api.cpp all expand this list.
//   sizes .. options
#define AMX_MODEL_INTS(X)                                                                                                  \
  X(nconst)                                                                                                                \
  X(nq) X(nv) X(nu) X(nb) X(njnt) X(ngc) X(npair) X(npc) /* collision pairs: [0, npc); [npc, npair - nbl) are the row groups of connect / weld equalities */ X(nbl) /* [npair - nbl, npair): ball-joint limit rows */ X(ncon) X(neq) X(nfri) /* friction-loss rows, between the equality and the limit rows */ X(nlim) X(nefc) X(nlevel) X(nM) X(nroot)                    \
  X(ldc) /* contact-row stride: longest pair dof list, odd */                                                              \
  X(ldm) /* row stride of M and H: >= nv, a multiple of 4 */                                                                \
  X(mrot) /* 1: row i of M stores chunk c at position (c + i) mod ldm/4 (fused path with an even chunk count) */          \
  X(fusedH) /* 1: Newton eliminates H in registers, no H buffer (G >= 32, no equality rows) */                            \
  X(mcomp) /* 1 (r05, the planner's dense form): M is stored compactly - rows [0, mnb0) of the leading chain every other dof descends from in full, every other row only the mwin 16-byte chunks that hold its entries on dofs >= mnb0 (its entries on the leading chain are read from those rows' columns: M is symmetric) */ X(mnb0) X(mwin) X(mbs) /* row stride of the whole rows: ldm, + 4 when ldm / 4 is even (sixteen-byte loads of consecutive rows then fall into different bank slots) */ X(jfr) /* 1 (dense form): e_J holds three rows per contact [3 ncon][ldc]: normal, mu * tangent 1, mu * tangent 2 */ \
  X(mfma2) /* 1: the contacts belong to exactly two pairs (two feet): J'DJ runs chain-compressed on the 4-block 16x16 matrix instruction, one block per (environment, pair) - amx_kernels.hip, newton_solve_pair2 */ X(p2_shl0) X(p2_shl1) /* row shift that takes a pair's rows from their lanes in the other 16-lane row to their dofs' lanes */ X(p2_banks) /* four 4-bit DPP bank masks: home adds of pair 0 | of pair 1 << 4 | cross adds into pair 1's row-0 dofs << 8 | into pair 0's row-1 dofs << 12 */ \
  X(nchol) X(njac) X(nhent) X(subtree_contig) /* 1: every body's descendants are the bodies right after it (field 6 of the body record carries the end) */ X(sub_iter) /* largest subtree (bodies, itself included) below the tree roots */ X(deep_free) /* 1: a free joint below the first tree level (the chain-walking velocity pass assumes none) */ X(max_dofmem) /* most contacts any one dof takes part in */ X(max_jntnum) /* most joints on one body */ X(max_bgrp) /* most dof groups on one body (c_bgrp) */ X(jac_one_sided) /* 1: every contact Jacobian entry belongs to the chain of geom2's body alone (geom1 on a body without dofs: the ground) */                                                                                               \
  X(nbs) /* contact-row slots per lane: 1, or 2 (r04; per-model kernels of Newton models with limit / joint-equality rows besides): rows gl and gl + G */ X(cx_first) X(ncx) X(cx_maxv) /* run of convex-convex pairs (fn 6), processed one after the other by the whole lane group */         \
  X(pcx_first) X(npcx) /* run of plane - convex pairs (<= 32 hull vertices) handled one lane per vertex; npcx = 0: none */ X(pcx_unit) /* lanes per pair in that routine: 16 when every hull of the run has at most 16 vertices, else 32 */   \
  X(iterations) X(ls_iterations) X(solver) X(disableflags)                                                                 \
  X(G)   /* lanes per environment: 16, 32 or 64 */                                                                         \
  X(wpb) /* wavefronts per workgroup (1..AMX_WAVES_PER_BLOCK), chosen to maximise resident waves per CU */                 \
  /* bases inside the constant image (float offsets) */                                                                    \
  X(c_body) X(c_jnt) X(c_dof) X(c_act) X(c_geom) X(c_pair) X(c_lim) X(c_eq) X(c_fri)                                                \
  X(c_qpos0) /* [nq] */ X(c_child) X(c_M) /* packed offset of M[i][j] | offset of M[j][i]<<11 | i<<22 | j<<27 */ X(c_dofact) X(c_pairdof) /* packed dof|sign<<8 */            \
  X(c_jac) /* packed contact|slot<<6|dof<<11|chain sign<<16|pair<<18 */                                                    \
  X(c_hent) /* [nhent] terms: packed dof_i|dof_j<<5|first contact row<<10|slot_i<<18|slot_j<<23 (dof_j <= dof_i) */         \
  X(c_dofmemadr) /* [nv+1] */ X(c_dofmem) /* terms: packed first contact row|slot<<8 */ X(c_conpair) /* [ncon] */           \
  X(c_vert) X(c_rootbody) X(c_slot) /* [ncon][32]: position of dof j in the contact's dof chain, -1 if absent */ X(c_zero) /* one all-zero matrix row (ldm floats, 16-byte aligned): what the lanes without a matrix row load (mcomp: 160 zeros - lane i reads its zero chunks at + 4 i, every lane of a sixteen-byte load's service group in a bank slot of its own) */ X(c_anc) /* [nb][max(nlevel-2,1)]: ancestors of a body, root first, itself and the world excluded, padded with 0 */ X(c_bgrp) /* [nb][max_bgrp]: the body's dof groups in order - first dof | count << 8 (1 or 3) | (1 << 12 unless the free joint's translations); 0: none */ X(c_vchdof) /* [npair][ldc]: dof of slot t of the pair's dof list, -1 past its end */ X(c_mrow) /* [nv] (mcomp): float offset of row i's storage inside M | first stored chunk << 12 | last << 16 */ X(c_p2) /* [32] (mfma2): per lane of an environment, the slot of the pair's dof list it feeds as A operand | as B operand << 8 (0xFF none, 0xFE the residual Jaref) */ X(c_prog) /* [AMX_WAVES_PER_BLOCK] zeros: in LDS the waves of a workgroup post their step counters here (fair sharing of a SIMD, amx_rollout_kernel) */            \
  /* bases inside one environment slab (float offsets from the slab start) */                                              \
  X(e_qpos) X(e_qvel) X(e_ctrl)                                                                                            \
  X(e_body) /* records [nb][BODY_E]: xpos 0-2 xquat 3-6 cinert 7-16 (then cfrc 7-12) crb 17-26 (then cvel 17-22 cacc 23-28) */ \
  X(e_dof)  /* records [nv][DOF_E]: cdof 0-5 | joint anchor+axis 6-11 in the joint's first dof, then fcrb, then Ma grad Mgrad search mv fcon | fsmooth 12 asmooth 13 qacc 14 warm 15 fbias 16 */ \
  X(e_com) X(e_actf)                                                                                                       \
  X(e_M) /* full symmetric [nv][ldm], 16-byte aligned rows (mcomp: [mnb0][ldm], then [nv - mnb0][4 mwin]) */ X(e_H) /* the same for H when it is stored; else the M*v staging vector */                                                       \
  X(e_Lb) /* packed lower-triangular Cholesky factor, only allocated for the CG solver */                                  \
  X(e_row) /* records [nefc][ROW_E]: Dact 0 frow 1 Jaref 2 (contact rows) */                                                                      \
  X(e_J) /* contact rows only, chain-compressed: [4*ncon][ldc] (jfr: [3*ncon][ldc]) */ X(e_vch) /* [npair][ldc]: the vector of the current J*v products gathered in the order of each pair's dof list (0 past its end) */                                                          \
  X(e_con) /* records [ncon][CON_E]: dist 0 pos 1-3 */ X(e_frame) /* [npair][9] */                                         \
  X(e_geom) /* records [ngc][GEO_E]: pos 0-2 mat 3-11 */                                                                   \
  X(e_cx) /* convex-convex scratch [6 cx_maxv + CX_TAIL], only when ncx > 0 */                                                                   \
  X(e_prof)                                                                                                                \
  X(env_floats) /* floats per environment slab */
#define AMX_MODEL_FLOATS(X) X(dt) X(gx) X(gy) X(gz) X(tolerance) X(ls_tolerance) X(impratio) X(scale) /* scale = meaninertia*max(1,nv) */

struct DevModel {
  const float *image;  // HBM copy of the constant image (ints stored bit-cast), nconst floats
#define X(f) int f;
  AMX_MODEL_INTS(X)
#undef X
#define X(f) float f;
  AMX_MODEL_FLOATS(X)
#undef X
};

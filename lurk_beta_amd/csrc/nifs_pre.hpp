// nifs_pre.hpp - the folding challenge r = RO(pp_digest, U1, U2, comm_T) in three stages (transcript.hip: NifsStages), for the folding
// context (step.hip): everything that does not depend on comm_T is absorbed - and the permutation it completes run - while the device
// still works on the step; behind commit(T) one permutation and one affine conversion remain.  Host code.
#pragma once
#include <cstddef>

namespace lurk {

struct NifsPre;
NifsPre* nifs_pre_begin(int curve, const void* pp_digest32, const void* comm_w1_jac96, const void* comm_e1_jac96, const void* u1_mont, const void* x1_mont,
                        size_t num_io);                                  // pp_digest, U1
void nifs_pre_fresh(NifsPre* p, const void* comm_w2_jac96, const void* x2_mont);  // U2
void nifs_pre_finish(NifsPre* p, const void* comm_t_jac96, void* r32_mont);       // comm_T, squeeze 128 bits -> r (Montgomery)
void nifs_pre_free(NifsPre* p);

}  // namespace lurk

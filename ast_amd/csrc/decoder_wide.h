// Persistent forward loop of the wide (H = A = 1024) decoder: decoder_wide.hip, driven by decoder.hip's forward loop.
#pragma once
#include "common.h"

namespace astk {

struct DecWideBuffers {     // slices of decoder.hip's DecPlan (layer 0)
  int32_t *TOK, *PRED;
  float *X0, *G, *C, *HR, *Q, *ALPHA, *CVH, *HT;
  float* PART;              // [B][nsplit][H + 4] + [B][ceil(V / 16)][2]   (decoder_wide_part_floats)
  unsigned* ctr;            // decoder_wide_ctr_words
};
bool decoder_wide_applicable(const astk_decoder_desc* d, int* nsplit_out, int* chunk_out);
size_t decoder_wide_part_floats(const astk_decoder_desc* d);
size_t decoder_wide_ctr_words(const astk_decoder_desc* d);
// decoder steps s0..s1 (inclusive) in one launch (normally all of them); a step that is not teacher-forced takes the argmax the kernel
// itself left in PRED for the step before (step s0: whatever PRED holds)
int decoder_wide_fwd_launch(const astk_decoder_desc* d, const astk_decoder_params* prm, const float* enc, const int32_t* y,
                            const int32_t* use_truth, const float* emb_mask, const float* rnn_mask, const DecWideBuffers& bf, int s0, int s1,
                            hipStream_t s);

struct DecWideBwdBuffers {
  const float *WcT, *WaT, *WlT, *WuT, *ALPHA, *CVH, *HT, *C;
  float *G, *DPRE, *DCVH, *DS, *DQ, *DHTOP, *DC0;
  float* scratch;           // decoder_wide_bwd_floats
  unsigned* ctr;            // decoder_wide_bwd_ctr_words
};
size_t decoder_wide_bwd_floats(const astk_decoder_desc* d);
size_t decoder_wide_bwd_ctr_words(const astk_decoder_desc* d);
// the whole reversed loop in one launch; expects d_pre's linear part (dlogits Wo) in DPRE and tanh' applied to its last step
int decoder_wide_bwd_launch(const astk_decoder_desc* d, const float* enc, const float* rnn_mask, const DecWideBwdBuffers& bf, hipStream_t s);

}  // namespace astk

// Internal constants shared by the contrastive kernels.
#pragma once
#include "common.h"

namespace ucd {
// Row granularity of the contrast matrix: the anchor segment and the teacher segment are each padded
// to a multiple of this many rows (padding rows carry label 255 and zero features).
constexpr int kPixTile = 128;
constexpr int kPadLabel = 255;

// fp16-operand loss path (pixcon_loss_f16.hip)
size_t pixcon16_workspace_bytes(int BHW);
int pixcon16_launch(const _Float16* ch16, const uint8_t* row_label, const _Float16* p16, int K,
                    const ucd_pixcon_meta* meta, int BHW, float temperature, int shift_pos, int use_prob,
                    float* loss_out, float* grad_a, int ldg, float* row_stats, void* workspace, size_t workspace_bytes,
                    hipStream_t s);
// planned, software-pipelined form of the same path (pixcon_loss_f16p.hip); eligible for T >= 0.06, at most 32 teacher
// classes and fewer than 1024 anchor blocks
bool pixcon16p_eligible(int BHW, float temperature, int use_prob, int K);
size_t pixcon16p_workspace_bytes(int BHW);
int pixcon16p_launch(const _Float16* ch16, const uint8_t* row_label, const _Float16* p16, int K,
                     const ucd_pixcon_meta* meta, int BHW, float temperature, int shift_pos, int use_prob,
                     float* loss_out, float* grad_a, int ldg, float* row_stats, void* workspace, size_t workspace_bytes,
                     hipStream_t s);
// loss_out[0] = sum(row_loss[0:A]) / n_valid, loss_out[1] = n_valid (pixcon_loss.hip)
void pixcon_launch_reduce(const float* row_loss, const ucd_pixcon_meta* meta, float* loss_out, hipStream_t s);
}  // namespace ucd

/* mirt_oracle_internal.h — helpers shared by the oracle's translation units. TEST INFRASTRUCTURE ONLY. */
#ifndef MIRT_ORACLE_INTERNAL_H
#define MIRT_ORACLE_INTERNAL_H

#include "mirt_oracle.h"

uint32_t mirt_params_out_rows_impl(const MirtParams* p);
uint32_t mirt_params_out_row_index_impl(const MirtParams* p, uint32_t i);
int      mirt_oracle_check(const MirtScene* scene, const MirtParams* params);
int      mirt_oracle_pick_threads(int n_threads);
void     mirt_oracle_stats_add(MirtStats* total, const MirtStats* part);

int mirt_oracle_render_parity(const MirtScene* scene, const MirtParams* params, uint8_t* out,
                              int n_threads, int variant, MirtStats* total);
int mirt_oracle_render_pt(const MirtScene* scene, const MirtParams* params, uint8_t* out_rgba8,
                          uint64_t* out_sums, int n_threads, MirtStats* total);

#endif

/* include/drnmf.h must be a plain C header: compiled with `gcc -std=c99 -Wall -Werror -pedantic
 * -fsyntax-only` by tests/test_host.py.  Taking the address of every export also checks that each
 * declaration is a complete prototype. */
#include "drnmf.h"

typedef void (*fn_t)(void);
#define REF(f) (fn_t)(f)

fn_t drnmf_all_exports[] = {
    REF(drnmf_version), REF(drnmf_create), REF(drnmf_destroy), REF(drnmf_last_error),
    REF(drnmf_params_bytes), REF(drnmf_prepare_params), REF(drnmf_cell_workspace_bytes),
    REF(drnmf_cell_forward), REF(drnmf_cell_forward_stateful), REF(drnmf_cell_forward_ista),
    REF(drnmf_cell_profile), REF(drnmf_dense_params_bytes), REF(drnmf_dense_prepare_params),
    REF(drnmf_dense_workspace_bytes), REF(drnmf_dense_cell_forward),
    REF(drnmf_dense_backward_workspace_bytes), REF(drnmf_dense_cell_backward), REF(drnmf_padded_f),
    REF(drnmf_head_forward), REF(drnmf_loss_head_workspace_bytes), REF(drnmf_loss_head_backward),
    REF(drnmf_snmf_cost_head_backward), REF(drnmf_cell_backward_workspace_bytes),
    REF(drnmf_cell_backward), REF(drnmf_cell_backward_stateful), REF(drnmf_cell_backward_ista), REF(drnmf_cell_backward_ista_stateful), REF(drnmf_adam_step), REF(drnmf_sumsq),
    REF(drnmf_adam_step_flat), REF(drnmf_adam_step_flat_counted), REF(drnmf_set_matrix_mode), REF(drnmf_get_matrix_mode),
    REF(drnmf_check_status), REF(drnmf_status_take_device),
    REF(drnmf_host_report_ring), REF(drnmf_reload_env), REF(drnmf_create_unbound), REF(drnmf_persist_admitted), REF(drnmf_persist_admit_reason),
    REF(drnmf_ista_workspace_bytes), REF(drnmf_ista_forward), REF(drnmf_mu_workspace_bytes),
    REF(drnmf_mu_forward), REF(drnmf_snmf_train_workspace_bytes), REF(drnmf_snmf_train_init),
    REF(drnmf_snmf_train_step), REF(drnmf_stft_frames), REF(drnmf_stft_mag), REF(drnmf_stft),
    REF(drnmf_istft_workspace_bytes), REF(drnmf_istft_masked), REF(drnmf_snr),
    REF(drnmf_sdr_workspace_bytes), REF(drnmf_sdr_corr), REF(drnmf_sdr_project),
    REF(drnmf_divide_a_by_aplusb), REF(drnmf_add), REF(drnmf_loss_forward_workspace_bytes),
    REF(drnmf_loss_forward), REF(drnmf_wav_int16_workspace_bytes), REF(drnmf_wav_int16),
    REF(drnmf_cell_backward_profile), REF(drnmf_comm_unique_id), REF(drnmf_comm_init),
    REF(drnmf_comm_destroy), REF(drnmf_comm_info), REF(drnmf_allreduce_grads),
    REF(drnmf_broadcast_params), REF(drnmf_cell_launches_per_frame),
    REF(drnmf_dense_cell_forward_dropout), REF(drnmf_dense_cell_backward_dropout),
    REF(drnmf_dense_cell_forward_dropout_stateful), REF(drnmf_dense_cell_backward_stateful),
};

/* Layout of the one struct that crosses the ABI in DEVICE memory, built by the host language
 * (dr-nmf_amd/ops.py adam_block_table: numpy dtype {u8 param, i8 flat_off, i4 count, i4 reserved}): 24 bytes,
 * no padding.  (C99 has no static assert: a negative array size fails the compilation.) */
#include <stddef.h>
typedef char drnmf_adam_block_is_24_bytes[sizeof(drnmf_adam_block_t) == 24 ? 1 : -1];
typedef char drnmf_adam_block_offsets[(offsetof(drnmf_adam_block_t, param) == 0 &&
                                       offsetof(drnmf_adam_block_t, flat_off) == 8 &&
                                       offsetof(drnmf_adam_block_t, count) == 16 &&
                                       offsetof(drnmf_adam_block_t, reserved) == 20) ? 1 : -1];
typedef char drnmf_cell_desc_is_12_int32[sizeof(drnmf_cell_desc_t) == 48 ? 1 : -1];

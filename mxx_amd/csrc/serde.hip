// serde.hip — compact wire format (coefficient-domain centred integers, bit-packed).
// Replaces the compact-bytes half of cuda/src/matrix/MatrixSerde.cu
// (cuda/include/matrix/MatrixSerde.cuh:35-58).
#include "common.h"
#include "modarith.h"

extern "C" int gpu_matrix_store_compact_bytes(GpuMatrix *mat, uint8_t *payload_out, size_t payload_capacity,
                                              uint16_t *out_max_coeff_bits, uint16_t *out_bytes_per_coeff,
                                              size_t *out_payload_len) {
    (void)mat; (void)payload_out; (void)payload_capacity; (void)out_max_coeff_bits; (void)out_bytes_per_coeff;
    (void)out_payload_len;
    return set_error("gpu_matrix_store_compact_bytes: not implemented yet");
}
extern "C" int gpu_matrix_load_compact_bytes(GpuMatrix *mat, const uint8_t *payload, size_t payload_len,
                                             uint16_t max_coeff_bits) {
    (void)mat; (void)payload; (void)payload_len; (void)max_coeff_bits;
    return set_error("gpu_matrix_load_compact_bytes: not implemented yet");
}
extern "C" int gpu_poly_store_compact_bytes(GpuMatrix *poly, uint8_t *payload_out, size_t payload_capacity,
                                            uint16_t *out_max_coeff_bits, uint16_t *out_bytes_per_coeff,
                                            size_t *out_payload_len) {
    return gpu_matrix_store_compact_bytes(poly, payload_out, payload_capacity, out_max_coeff_bits, out_bytes_per_coeff,
                                          out_payload_len);
}
extern "C" int gpu_poly_load_compact_bytes(GpuMatrix *poly, const uint8_t *payload, size_t payload_len,
                                           uint16_t max_coeff_bits) {
    return gpu_matrix_load_compact_bytes(poly, payload, payload_len, max_coeff_bits);
}
extern "C" int gpupoly_matrix_mul_decompose(GpuMatrix *out, const GpuMatrix *lhs, const GpuMatrix *rhs,
                                            uint32_t base_bits) {
    (void)out; (void)lhs; (void)rhs; (void)base_bits;
    return set_error("gpupoly_matrix_mul_decompose: not implemented yet");
}

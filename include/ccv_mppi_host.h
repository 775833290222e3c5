/*
 * ccv_mppi_host.h -- host-side prologue of the MPPI hot path (plain C ABI, no device work).
 *
 * These are the O(path)+O(H) pieces that stay on the host in the MI355X design and the
 * synthetic reference-path generators that define the benchmark inputs:
 *
 *   ccv_mppi_calc_ref_path   get_CurrentIndex() + calc_RefPath()   src/diff_drive_mppi.cpp:126-140,156-181
 *                                                                  src/steering_diff_drive_mppi.cpp:142-156,172-197
 *                                                                  src/full_body_mppi.cpp:335-349,365-392
 *   ccv_mppi_path_cosine     ReferencePathCreater::run(), "sin" branch   src/reference_path_creator.cpp:37-56
 *   ccv_mppi_path_dkan       DkanPathCreater::run()                      src/dkan_path_creator.cpp:11-35,37-51,62-64
 *   ccv_mppi_plant_step      the Euler model of predict_NextState() applied to the real pose (closed-loop harness)
 */
#ifndef CCV_MPPI_HOST_H_
#define CCV_MPPI_HOST_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Fills x_ref[H], y_ref[H], yaw_ref[H-1] (yaw_ref[H-1] is left untouched, as in the reference) and returns
 * current_index_ (>= 0), or a negative CCV_MPPI_ERR_* code.  The window index is the truncation of
 * current_index + i * v_ref * dt / resolution (dd:160-163) and the node takes dt from its clock (dd:346-348): a dt that
 * is negative or not finite, or a stride that is negative / not finite, would index before path_[0] (undefined behaviour
 * in the reference) and is refused with CCV_MPPI_ERR_INVALID_ARG -- nothing is written to x_ref / y_ref / yaw_ref then, and
 * the caller must not iterate against the stale window (MPPIBase::run_once() refuses the tick).  dt == 0 is defined in the
 * reference (stride 0: H copies of the nearest pose, as with v_ref == 0) and admitted. */
int ccv_mppi_calc_ref_path(const double* path_x, const double* path_y, int32_t n_path, double cur_x, double cur_y,
                           double v_ref, double dt, double resolution, int32_t horizon, double* x_ref, double* y_ref,
                           double* yaw_ref);

/* y = A1 cos(2 pi w1 s + d1) + A2 cos(..) + A3 cos(..) + init_y - (A1+A2+A3), x = init_x + s, s += resolution while
 * s < course_length.  A/omega/delta are arrays of 3.  Returns the number of poses written (<= cap). */
int ccv_mppi_path_cosine(const double* A, const double* omega, const double* delta, double resolution,
                         double course_length, double init_x, double init_y, double* path_x, double* path_y,
                         int32_t cap);

/* (0,0) -> (17.7,0) -> (17.7,8) -> (0,8) sampled every `resolution` metres. Returns the number of poses written. */
int ccv_mppi_path_dkan(double resolution, double* path_x, double* path_y, int32_t cap);

/* state (x, y, yaw[, roll, pitch]) advanced one Euler step with the controls u (model's control order).  sin / cos of
 * the heading are the specified polynomial evaluation the device uses (<= 1 ulp from libm), so that this and the resident
 * loop's plant (ccv_mppi_resident_step_enqueue, ccv_mppi.h) give the same bits; |heading| <= 1e5.  yaw / roll / pitch beyond
 * +-1e4 rad are taken modulo 2 pi after the step (the node reads them from tf in [-pi, pi]); the device plant does the same. */
int ccv_mppi_plant_step(int32_t model, double* state, const double* u, double dt);

#ifdef __cplusplus
}
#endif
#endif /* CCV_MPPI_HOST_H_ */

#!/bin/bash
# dev helper: timing build (tools/mk_fwd_timing.sh) + a hook that exports / overrides the workgroup schedule of the one-call
# forward (ur_dev_set_order(out, in)), for tools/fwd_order_oracle.py.  Output abx/libf_ord.so
NAME=f_ord bash "$(dirname "$0")/mk_fwd_timing.sh" \
 '    if (rc) return rc;@@  } else {@@    // large meshes: the per-face kernels=>    if (rc) return rc;@@    if (g_dev_order_out) (void)hipMemcpyAsync(g_dev_order_out, order, (size_t)N * nbins * 4, hipMemcpyDeviceToDevice, st);@@    if (g_dev_order_in) (void)hipMemcpyAsync(order, g_dev_order_in, (size_t)N * nbins * 4, hipMemcpyDeviceToDevice, st);@@  } else {@@    // large meshes: the per-face kernels' \
 '// 64-pixel groups (one per wave) of an image: 4 per 16x16 bin=>int* g_dev_order_out = nullptr;@@int* g_dev_order_in = nullptr;@@extern "C" void ur_dev_set_order(void* out, void* in) { g_dev_order_out = (int*)out; g_dev_order_in = (int*)in; }@@// 64-pixel groups (one per wave) of an image: 4 per 16x16 bin'

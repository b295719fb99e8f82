// kernels_fused3d.hip -- placeholder TU, filled in below

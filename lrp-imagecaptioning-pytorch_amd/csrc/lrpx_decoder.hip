// decoder kernels (filled in next)

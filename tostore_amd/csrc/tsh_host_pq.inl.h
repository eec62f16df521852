// tsh_host_pq.inl.h -- write-path helpers: PQ codebook training and batch encode entry points
// Part of the single translation unit tsh_lib.hip (textually included there; not compiled alone).

// ---- N4: batch PQ encode of resident rows -------------------------------------------
extern "C" int32_t tsh_pq_train(int32_t device, const float *samples, int64_t n, int32_t dim, int32_t subspaces,
                                int32_t centroids, int32_t iterations, const int32_t *init_index,
                                float *out_codebook) {
  if (!samples || !init_index || !out_codebook) return set_err(TSH_E_BAD_ARG, "NULL pointer");
  if (n < 1 || n > (1 << 24) || dim < 1) return set_err(TSH_E_BAD_ARG, "samples %lld x %d out of range", (long long)n, dim);
  if (subspaces < 1 || subspaces > dim || centroids < 1 || centroids > 256 || iterations < 0)
    return set_err(TSH_E_BAD_ARG, "subspaces %d / centroids %d / iterations %d out of range", subspaces, centroids,
                   iterations);
  const int32_t sd = dim / subspaces;
  if (sd < 1 || sd > 64) return set_err(TSH_E_BAD_ARG, "sub-space width %d outside [1,64]", sd);
  for (int64_t i = 0; i < (int64_t)subspaces * centroids; ++i)
    if (init_index[i] < 0 || init_index[i] >= n) return set_err(TSH_E_BAD_ARG, "init_index[%lld] outside the samples", (long long)i);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return set_err(TSH_E_NO_DEVICE, "no HIP device");
  if (device < 0 || device >= ndev) return set_err(TSH_E_BAD_ARG, "device %d of %d", device, ndev);
  HIPCHK(hipSetDevice(device));
  const size_t cb_elems = (size_t)subspaces * centroids * sd;
  float *d_samples = nullptr, *d_cent = nullptr, *d_norms = nullptr;
  int32_t *d_ints = nullptr;  // assign | init | active | changed
  const size_t n_ints = (size_t)subspaces * n + (size_t)subspaces * centroids + 2 * (size_t)subspaces;
  hipStream_t st = nullptr;
  hipError_t e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipMalloc(&d_samples, (size_t)n * dim * sizeof(float));
  if (e == hipSuccess) e = hipMalloc(&d_cent, cb_elems * sizeof(float));
  if (e == hipSuccess) e = hipMalloc(&d_norms, (size_t)subspaces * centroids * sizeof(float));
  if (e == hipSuccess) e = hipMalloc(&d_ints, n_ints * sizeof(int32_t));
  if (e == hipSuccess) e = hipMemcpyAsync(d_samples, samples, (size_t)n * dim * sizeof(float), hipMemcpyHostToDevice, st);
  if (e == hipSuccess) {
    PqTrainArgs a{};
    a.samples = d_samples;
    a.centroids = d_cent;
    a.norms = d_norms;
    a.assign = d_ints;
    int32_t *d_init = d_ints + (size_t)subspaces * n;
    a.init_idx = d_init;
    a.active = d_init + (size_t)subspaces * centroids;
    a.changed = a.active + subspaces;
    a.n = (int32_t)n;
    a.dim = dim;
    a.subspaces = subspaces;
    a.k = centroids;
    a.sub_dim = sd;
    e = hipMemcpyAsync(d_init, init_index, (size_t)subspaces * centroids * sizeof(int32_t), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
      pq_train_init_kernel<<<dim3(centroids, subspaces), 64, 0, st>>>(a);
      const dim3 ga((unsigned)((n + 255) / 256), subspaces);
      for (int it = 0; it < iterations; ++it) {
        pq_train_norms_kernel<<<subspaces, 256, 0, st>>>(a);
        if (sd == 8) pq_train_assign_kernel<8, true><<<ga, 256, 0, st>>>(a);
        else if (sd == 4) pq_train_assign_kernel<4, true><<<ga, 256, 0, st>>>(a);
        else if (sd == 16) pq_train_assign_kernel<16, true><<<ga, 256, 0, st>>>(a);
        else if (sd % 4 == 0) pq_train_assign_kernel<0, true><<<ga, 256, 0, st>>>(a);
        else pq_train_assign_kernel<0, false><<<ga, 256, 0, st>>>(a);
        pq_train_update_kernel<<<dim3(centroids, subspaces), 64, 0, st>>>(a);
        pq_train_flag_kernel<<<(subspaces + 255) / 256, 256, 0, st>>>(a);
      }
      e = hipMemcpyAsync(out_codebook, d_cent, cb_elems * sizeof(float), hipMemcpyDeviceToHost, st);
    }
  }
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e == hipSuccess) e = hipGetLastError();
  if (d_samples) hipFree(d_samples);
  if (d_cent) hipFree(d_cent);
  if (d_norms) hipFree(d_norms);
  if (d_ints) hipFree(d_ints);
  if (st) hipStreamDestroy(st);
  if (e != hipSuccess) return set_err(TSH_E_HIP, "pq_train: %s", hipGetErrorString(e));
  return TSH_OK;
}

extern "C" int32_t tsh_index_pq_encode(tsh_index *idx, int64_t first_row_id, int64_t n_rows, const float *codebook,
                                       int32_t subspaces, int32_t centroids, uint8_t *out_codes) {
  if (!idx || !codebook || !out_codes) return set_err(TSH_E_BAD_ARG, "NULL pointer");
  if (n_rows < 0 || first_row_id < 0) return set_err(TSH_E_BAD_ARG, "negative row range");
  if (subspaces < 1 || subspaces > idx->dim || centroids < 1 || centroids > 256)
    return set_err(TSH_E_BAD_ARG, "subspaces %d / centroids %d out of range", subspaces, centroids);
  const int32_t sub_dim = idx->dim / subspaces;  // PqCodebook: dimensions = subspaces * subDimensions
  if (sub_dim < 1 || sub_dim > 64) return set_err(TSH_E_BAD_ARG, "sub-space width %d outside [1,64]", sub_dim);
  if (n_rows == 0) return TSH_OK;
  const size_t cb_elems = (size_t)subspaces * centroids * sub_dim;
  std::vector<double> cb64(cb_elems);
  for (size_t i = 0; i < cb_elems; ++i) cb64[i] = (double)codebook[i];  // exact widening
  int64_t done = 0;
  while (done < n_rows) {
    const int64_t gid = first_row_id + done;
    Shard *s = shard_for_row(idx, gid);
    std::shared_lock<RwLock> sl = share(idx, s);
    const int64_t local = gid - s->row_base;
    if (local < 0 || local >= s->rows)
      return set_err(TSH_E_BAD_ARG, "row %lld is not resident", (long long)gid);
    const int64_t take = std::min(n_rows - done, s->rows - local);
    HIPCHK(hipSetDevice(s->device));
    double *d_cb = nullptr;
    uint8_t *d_codes = nullptr;
    HIPCHK(hipMalloc(&d_cb, cb_elems * sizeof(double)));
    hipError_t e = hipMalloc(&d_codes, (size_t)take * subspaces);
    hipStream_t st = s->ingest_stream;
    if (e == hipSuccess) e = hipMemcpyAsync(d_cb, cb64.data(), cb_elems * sizeof(double), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
      PqEncodeArgs a{};
      a.rows = s->d_rows;
      a.codebook = d_cb;
      a.codes = d_codes;
      a.ld = s->ld;
      a.first = local;
      a.n = take;
      a.subspaces = subspaces;
      a.centroids = centroids;
      a.sub_dim = sub_dim;
      const unsigned grid = (unsigned)((take + 255) / 256);
      if (sub_dim == 8) pq_encode_kernel<8><<<grid, 256, 0, st>>>(a);
      else if (sub_dim == 4) pq_encode_kernel<4><<<grid, 256, 0, st>>>(a);
      else if (sub_dim == 16) pq_encode_kernel<16><<<grid, 256, 0, st>>>(a);
      else pq_encode_kernel<0><<<grid, 256, 0, st>>>(a);
      e = hipMemcpyAsync(out_codes + (size_t)done * subspaces, d_codes, (size_t)take * subspaces,
                         hipMemcpyDeviceToHost, st);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e == hipSuccess) e = hipGetLastError();
    hipFree(d_cb);
    if (d_codes) hipFree(d_codes);
    if (e != hipSuccess) return set_err(TSH_E_HIP, "pq_encode: %s", hipGetErrorString(e));
    done += take;
  }
  return TSH_OK;
}


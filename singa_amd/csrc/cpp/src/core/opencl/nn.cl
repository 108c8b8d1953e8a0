// Convolution and pooling of the OpenCL device (float32, NCHW), as gathers:
// one work-item per output element of each pass, so no pass needs atomics
// or a zero-filled destination.  Role: reference src/model/layer/
// opencl_convolution.cc + im2col.cl, opencl_pooling.cc + pooling.cl.

typedef struct {
  int N, C, H, W;
  int K, R, S;
  int P, Q;
  int stride_h, stride_w, pad_h, pad_w;
  int groups;
} ConvShape;

typedef struct {
  int N, C, H, W, P, Q;
  int kh, kw, sh, sw, ph, pw;
  int is_max;
} PoolShape;

__kernel void k_conv_fwd(ConvShape g, __global const float *x,
                         __global const float *w, __global const float *bias,
                         int has_bias, __global float *y) {
  const int Cg = g.C / g.groups, Kg = g.K / g.groups;
  const long total = (long)g.N * g.K * g.P * g.Q;
  for (long i = get_global_id(0); i < total; i += get_global_size(0)) {
    const int q = i % g.Q;
    const int p = (i / g.Q) % g.P;
    const int k = (i / ((long)g.Q * g.P)) % g.K;
    const int n = i / ((long)g.Q * g.P * g.K);
    const int c0 = (k / Kg) * Cg;
    float acc = has_bias ? bias[k] : 0.f;
    for (int c = 0; c < Cg; ++c)
      for (int r = 0; r < g.R; ++r) {
        const int h = p * g.stride_h - g.pad_h + r;
        if (h < 0 || h >= g.H) continue;
        for (int s = 0; s < g.S; ++s) {
          const int ww = q * g.stride_w - g.pad_w + s;
          if (ww < 0 || ww >= g.W) continue;
          acc = fma(x[(((long)n * g.C + c0 + c) * g.H + h) * g.W + ww],
                    w[(((long)k * Cg + c) * g.R + r) * g.S + s], acc);
        }
      }
    y[i] = acc;
  }
}

// dx[n, c, h, w] = sum over (k, r, s) with an output position p, q such that
// p * stride - pad + r == h (likewise for w)
__kernel void k_conv_bwd_data(ConvShape g, __global const float *dy,
                              __global const float *w, __global float *dx) {
  const int Cg = g.C / g.groups, Kg = g.K / g.groups;
  const long total = (long)g.N * g.C * g.H * g.W;
  for (long i = get_global_id(0); i < total; i += get_global_size(0)) {
    const int ww = i % g.W;
    const int h = (i / g.W) % g.H;
    const int c = (i / ((long)g.W * g.H)) % g.C;
    const int n = i / ((long)g.W * g.H * g.C);
    const int grp = c / Cg, cl = c % Cg;
    float acc = 0.f;
    for (int r = 0; r < g.R; ++r) {
      const int ph = h + g.pad_h - r;
      if (ph < 0 || ph % g.stride_h) continue;
      const int p = ph / g.stride_h;
      if (p >= g.P) continue;
      for (int s = 0; s < g.S; ++s) {
        const int qw = ww + g.pad_w - s;
        if (qw < 0 || qw % g.stride_w) continue;
        const int q = qw / g.stride_w;
        if (q >= g.Q) continue;
        for (int kk = 0; kk < Kg; ++kk) {
          const int k = grp * Kg + kk;
          acc = fma(dy[(((long)n * g.K + k) * g.P + p) * g.Q + q],
                    w[(((long)k * Cg + cl) * g.R + r) * g.S + s], acc);
        }
      }
    }
    dx[i] = acc;
  }
}

// dw[k, c, r, s]: one work-group per filter tap, its items share the
// (n, p, q) positions and meet in local memory
__kernel void k_conv_bwd_weight(ConvShape g, __global const float *dy,
                                __global const float *x,
                                __global float *dw) {
  __local float buf[256];
  const int Cg = g.C / g.groups, Kg = g.K / g.groups;
  const long taps = (long)g.K * Cg * g.R * g.S;
  const long positions = (long)g.N * g.P * g.Q;
  for (long t = get_group_id(0); t < taps; t += get_num_groups(0)) {
    const int s = t % g.S;
    const int r = (t / g.S) % g.R;
    const int c = (t / ((long)g.S * g.R)) % Cg;
    const int k = t / ((long)g.S * g.R * Cg);
    const int cin = (k / Kg) * Cg + c;
    float acc = 0.f;
    for (long j = get_local_id(0); j < positions; j += get_local_size(0)) {
      const int q = j % g.Q;
      const int p = (j / g.Q) % g.P;
      const int n = j / ((long)g.Q * g.P);
      const int h = p * g.stride_h - g.pad_h + r;
      const int ww = q * g.stride_w - g.pad_w + s;
      if (h < 0 || h >= g.H || ww < 0 || ww >= g.W) continue;
      acc = fma(dy[(((long)n * g.K + k) * g.P + p) * g.Q + q],
                x[(((long)n * g.C + cin) * g.H + h) * g.W + ww], acc);
    }
    const int l = get_local_id(0);
    buf[l] = acc;
    barrier(CLK_LOCAL_MEM_FENCE);
    for (int span = get_local_size(0) / 2; span > 0; span >>= 1) {
      if (l < span) buf[l] += buf[l + span];
      barrier(CLK_LOCAL_MEM_FENCE);
    }
    if (l == 0) dw[t] = buf[0];
    barrier(CLK_LOCAL_MEM_FENCE);
  }
}

// db[k] = sum over (n, p, q) of dy; one work-group per channel
__kernel void k_conv_bwd_bias(__global const float *dy, __global float *db,
                              int N, int K, int PQ) {
  __local float buf[256];
  for (int k = get_group_id(0); k < K; k += get_num_groups(0)) {
    float acc = 0.f;
    const long count = (long)N * PQ;
    for (long j = get_local_id(0); j < count; j += get_local_size(0)) {
      const int n = j / PQ, i = j % PQ;
      acc += dy[((long)n * K + k) * PQ + i];
    }
    const int l = get_local_id(0);
    buf[l] = acc;
    barrier(CLK_LOCAL_MEM_FENCE);
    for (int span = get_local_size(0) / 2; span > 0; span >>= 1) {
      if (l < span) buf[l] += buf[l + span];
      barrier(CLK_LOCAL_MEM_FENCE);
    }
    if (l == 0) db[k] = buf[0];
    barrier(CLK_LOCAL_MEM_FENCE);
  }
}

__kernel void k_pool_fwd(PoolShape g, __global const float *x,
                         __global float *y) {
  const long total = (long)g.N * g.C * g.P * g.Q;
  for (long i = get_global_id(0); i < total; i += get_global_size(0)) {
    const int ow = i % g.Q;
    const int oh = (i / g.Q) % g.P;
    const long nc = i / ((long)g.Q * g.P);
    float acc = g.is_max ? -FLT_MAX : 0.f;
    int cnt = 0;
    for (int r = 0; r < g.kh; ++r) {
      const int h = oh * g.sh - g.ph + r;
      if (h < 0 || h >= g.H) continue;
      for (int s = 0; s < g.kw; ++s) {
        const int w = ow * g.sw - g.pw + s;
        if (w < 0 || w >= g.W) continue;
        const float v = x[(nc * g.H + h) * g.W + w];
        acc = g.is_max ? fmax(acc, v) : acc + v;
        ++cnt;
      }
    }
    y[i] = g.is_max ? acc : acc / (float)max(cnt, 1);
  }
}

// Gather form of the pooling gradient: an input cell collects from every
// window that covers it.  Max pooling credits the FIRST cell (row-major
// within the window) that equals the window's maximum, as the host code does.
__kernel void k_pool_bwd(PoolShape g, __global const float *dy,
                         __global const float *x, __global const float *y,
                         __global float *dx) {
  const long total = (long)g.N * g.C * g.H * g.W;
  for (long i = get_global_id(0); i < total; i += get_global_size(0)) {
    const int w = i % g.W;
    const int h = (i / g.W) % g.H;
    const long nc = i / ((long)g.W * g.H);
    const float xv = x[i];
    float acc = 0.f;
    for (int oh = 0; oh < g.P; ++oh) {
      const int r = h - (oh * g.sh - g.ph);
      if (r < 0 || r >= g.kh) continue;
      for (int ow = 0; ow < g.Q; ++ow) {
        const int s = w - (ow * g.sw - g.pw);
        if (s < 0 || s >= g.kw) continue;
        const long yo = (nc * g.P + oh) * g.Q + ow;
        if (g.is_max) {
          if (xv != y[yo]) continue;
          // an earlier cell of the window with the same value takes it
          int earlier = 0;
          for (int rr = 0; rr <= r && !earlier; ++rr) {
            const int hh = oh * g.sh - g.ph + rr;
            if (hh < 0 || hh >= g.H) continue;
            const int s_end = rr == r ? s : g.kw;
            for (int ss = 0; ss < s_end; ++ss) {
              const int wc = ow * g.sw - g.pw + ss;
              if (wc < 0 || wc >= g.W) continue;
              if (x[(nc * g.H + hh) * g.W + wc] == xv) { earlier = 1; break; }
            }
          }
          if (!earlier) acc += dy[yo];
        } else {
          int cnt = 0;
          for (int rr = 0; rr < g.kh; ++rr) {
            const int hh = oh * g.sh - g.ph + rr;
            if (hh < 0 || hh >= g.H) continue;
            for (int ss = 0; ss < g.kw; ++ss) {
              const int wc = ow * g.sw - g.pw + ss;
              if (wc >= 0 && wc < g.W) ++cnt;
            }
          }
          acc += dy[yo] / (float)max(cnt, 1);
        }
      }
    }
    dx[i] = acc;
  }
}

// ---- batch normalisation ---------------------------------------------------------
// One work-group per channel: two-pass statistics (mean first, then the
// centred second moment), so the variance does not suffer from cancellation.
inline float group_sum(float v, __local float *buf) {
  const int l = get_local_id(0);
  buf[l] = v;
  barrier(CLK_LOCAL_MEM_FENCE);
  for (int span = get_local_size(0) / 2; span > 0; span >>= 1) {
    if (l < span) buf[l] += buf[l + span];
    barrier(CLK_LOCAL_MEM_FENCE);
  }
  const float r = buf[0];
  barrier(CLK_LOCAL_MEM_FENCE);
  return r;
}

__kernel void k_bn_fwd_train(__global const float *x,
                             __global const float *scale,
                             __global const float *bias,
                             __global float *rmean, __global float *rvar,
                             float factor, float eps, __global float *y,
                             __global float *mean_out,
                             __global float *invstd_out, int N, int C,
                             int HW) {
  __local float buf[256];
  const long count = (long)N * HW;
  for (int c = get_group_id(0); c < C; c += get_num_groups(0)) {
    float s = 0.f;
    for (long j = get_local_id(0); j < count; j += get_local_size(0))
      s += x[((j / HW) * C + c) * HW + j % HW];
    const float mean = group_sum(s, buf) / (float)count;
    float ss = 0.f;
    for (long j = get_local_id(0); j < count; j += get_local_size(0)) {
      const float d = x[((j / HW) * C + c) * HW + j % HW] - mean;
      ss += d * d;
    }
    const float var = group_sum(ss, buf) / (float)count;
    const float invstd = rsqrt(var + eps);
    if (get_local_id(0) == 0) {
      mean_out[c] = mean;
      invstd_out[c] = invstd;
      const float unbiased =
          count > 1 ? var * (float)count / (float)(count - 1) : var;
      rmean[c] = (1.f - factor) * rmean[c] + factor * mean;
      rvar[c] = (1.f - factor) * rvar[c] + factor * unbiased;
    }
    const float g = scale[c] * invstd, b = bias[c] - mean * g;
    for (long j = get_local_id(0); j < count; j += get_local_size(0)) {
      const long o = ((j / HW) * C + c) * HW + j % HW;
      y[o] = fma(x[o], g, b);
    }
  }
}

__kernel void k_bn_fwd_infer(__global const float *x,
                             __global const float *scale,
                             __global const float *bias,
                             __global const float *rmean,
                             __global const float *rvar, float eps,
                             __global float *y, long total, int C, int HW) {
  for (long i = get_global_id(0); i < total; i += get_global_size(0)) {
    const int c = (i / HW) % C;
    const float g = scale[c] * rsqrt(rvar[c] + eps);
    y[i] = fma(x[i] - rmean[c], g, bias[c]);
  }
}

__kernel void k_bn_bwd(__global const float *dy, __global const float *x,
                       __global const float *scale,
                       __global const float *mean,
                       __global const float *invstd, __global float *dx,
                       __global float *dscale, __global float *dbias, int N,
                       int C, int HW) {
  __local float buf[256];
  const long count = (long)N * HW;
  for (int c = get_group_id(0); c < C; c += get_num_groups(0)) {
    const float m = mean[c], is = invstd[c];
    float sg = 0.f, sgx = 0.f;
    for (long j = get_local_id(0); j < count; j += get_local_size(0)) {
      const long o = ((j / HW) * C + c) * HW + j % HW;
      sg += dy[o];
      sgx += dy[o] * (x[o] - m) * is;
    }
    sg = group_sum(sg, buf);
    sgx = group_sum(sgx, buf);
    if (get_local_id(0) == 0) {
      dbias[c] = sg;
      dscale[c] = sgx;
    }
    const float k = scale[c] * is, inv = 1.f / (float)count;
    for (long j = get_local_id(0); j < count; j += get_local_size(0)) {
      const long o = ((j / HW) * C + c) * HW + j % HW;
      const float xh = (x[o] - m) * is;
      dx[o] = k * (dy[o] - sg * inv - xh * sgx * inv);
    }
  }
}

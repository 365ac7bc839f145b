// Epilogue of the Winograd F(2x2,3x3) convolution kernels: A^T M A of the 16 component accumulators a lane holds for ONE 2x2 tile and
// 4 x NIW out-channels, then the fused point-wise tail (bias, LeakyReLU, PixelNorm, AvgPool2d, tile masks, un-pool, fade-in blends).
// Textually included INSIDE a kernel body (no include guard): wino3x3_mfma (wino3x3.hip) and wino3x3_strip (wino_strip.hip).  Names it
// expects in scope: a (WinoArgs), acc[16][NIW], NIW, WC, TPB, lane, col, rq, wt, wc, ct0, HW, Ht, Wt, ebx, eby, en0 (the tile block
// of the accumulators), red (LDS, PixelNorm partial sums across the WC wave groups; unused when WC == 1).
  // ---------------------------------------------------------------- epilogue: A^T M A, then the fused point-wise tail
  const float slope_eff = (a.flags & MG_CONV_LRELU) ? a.slope : 1.0f;  // branch-free LeakyReLU switch
  const int tl = wt * 16 + col;
  const int txl = tl & (a.TBW - 1);
  const int tyl = (tl >> a.lgTBW) & (a.TBH - 1);
  const int nl = tl >> (a.lgTBW + a.lgTBH);
  const int n = en0 + nl, TY = eby * a.TBH + tyl, TX = ebx * a.TBW + txl;
  const bool tok = (n < a.N) && (TY < Ht) && (TX < Wt);
  const int oc0 = (ct0 + wc * NIW) * 16 + rq * 4;  // out-channel of (ni, g) = oc0 + ni*16 + g
  const size_t pix0 = ((size_t)n * a.Cout * a.H + 2 * TY) * a.W + 2 * TX;  // + oc*H*W + i*W
  const size_t pp0 = ((size_t)n * a.Cout * Ht + TY) * Wt + TX;            // + oc*Ht*Wt   (pooled tensor)

  // A^T M A + bias + LeakyReLU of out-channel tile ni -> on[g][2*i + j: pixel (i, j) of the lane's 2x2 tile].  The accumulator
  // of Winograd component (xi, nu) is acc[4*xi + slot(nu)] with slots [nu0, nu3, nu1, nu2] (see store_chunk); all math runs on
  // the f32x4 accumulators (g = 4 out-channels) so it packs.
  auto transform = [&](int ni, float (&on)[4][4], auto act_) __attribute__((always_inline)) {
    constexpr bool ACT = decltype(act_)::value;  // bias + LeakyReLU present (forward paths)
    f32x4 bv4;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int oc = oc0 + ni * 16 + g;
      bv4[g] = (a.bias != nullptr && oc < a.Cout) ? a.bias[oc] : 0.f;
    }
    constexpr int SL[4] = {0, 2, 3, 1};  // slot of column nu
    f32x4 s0[4], s1[4];
#pragma unroll
    for (int nu = 0; nu < 4; ++nu) {
      const f32x4 m0 = acc[SL[nu]][ni], m1 = acc[4 + SL[nu]][ni], m2 = acc[8 + SL[nu]][ni], m3 = acc[12 + SL[nu]][ni];
      s0[nu] = (m0 + m1) + m2;
      s1[nu] = (m1 - m2) - m3;
    }
    f32x4 r4[4];
    r4[0] = (s0[0] + s0[1]) + s0[2];
    r4[1] = (s0[1] - s0[2]) - s0[3];
    r4[2] = (s1[0] + s1[1]) + s1[2];
    r4[3] = (s1[1] - s1[2]) - s1[3];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if constexpr (ACT) {
        const f32x4 v = r4[q] + bv4;
        const f32x4 w = v * slope_eff;  // 0 < slope <= 1: leaky_relu(v) == max(v, slope*v)
#pragma unroll
        for (int g = 0; g < 4; ++g) on[g][q] = fmaxf(v[g], w[g]);
      } else {
#pragma unroll
        for (int g = 0; g < 4; ++g) on[g][q] = r4[q][g];
      }
    }
  };
  // one specialised copy of the store loop per epilogue kind, selected once (no flag tests inside the unrolled loops).
  // Stores are issue-bound per instruction, so lane pairs (tiles 2k, 2k+1 of a row) first trade half of their 2x2 outputs
  // through DPP: the even lane ends up with image row 2TY of both tiles, the odd lane with row 2TY+1 -- 16 contiguous bytes
  // per lane, 128 per 8 lanes -- and each (tile, out-channel) costs one dwordx4 store (and mask load) instead of two dwordx2.
  const bool odd = (lane & 1) != 0;
  const bool wide = (a.TBW >= 2) && ((Wt & 1) == 0);  // tile pairs exist and share validity
  auto swap1 = [&](float x) __attribute__((always_inline)) {  // value of lane ^ 1
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xf, 0xf, false));
  };
  const size_t pixw = odd ? pix0 - 2 + a.W : pix0;  // even: row 2TY of the pair; odd: row 2TY+1 of the pair
  auto store_tile = [&](int ni, float (&on)[4][4], const float (&rnv)[4], auto mask_, auto pn_, auto pool_, auto hasy_,
                        auto mout_, auto blend_) __attribute__((always_inline)) {
    constexpr int MASKK = decltype(mask_)::value;  // 0: none, 1: aux = fp32 activations (N,Cout,H,W), 2: a.mi = tile bytes
    constexpr bool MASK = MASKK == 1, PN = decltype(pn_)::value, POOL = decltype(pool_)::value,
                   HASY = decltype(hasy_)::value, MOUT = decltype(mout_)::value, BLEND = decltype(blend_)::value;
    float bca = 1.f, bcb = 0.f;
    if constexpr (BLEND) { bca = a.coef[0]; bcb = a.coef[1]; }
    if constexpr (MASKK == 2 || MOUT) {  // one byte per 2x2 tile and out-channel, bit 2i+j <-> pixel (i, j): the lane's own tile
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int oc = oc0 + ni * 16 + g;
        const bool act = tok && oc < a.Cout;
        const size_t ti = pp0 + (size_t)oc * (Ht * Wt);
        if constexpr (MASKK == 2) {
          const unsigned mb = act ? (unsigned)a.mi[ti] : 15u;
#pragma unroll
          for (int q = 0; q < 4; ++q) on[g][q] *= ((mb >> q) & 1u) ? 1.f : a.slope;
        }
        if constexpr (MOUT) {
          const unsigned mb = ((mg_pos_bit(on[g][0]) | (mg_pos_bit(on[g][1]) << 1)) | (mg_pos_bit(on[g][2]) << 2)) | (mg_pos_bit(on[g][3]) << 3);
          if (act) a.mo[ti] = (unsigned char)mb;
        }
      }
    }
    if (wide) {
      f32x4 rnw = f32x4{1.f, 1.f, 1.f, 1.f};
      if constexpr (PN) {
        const float r0 = swap1(odd ? rnv[0] : rnv[2]), r1 = swap1(odd ? rnv[1] : rnv[3]);
        rnw = odd ? f32x4{r0, r1, rnv[2], rnv[3]} : f32x4{rnv[0], rnv[1], r0, r1};
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int oc = oc0 + ni * 16 + g;
        const bool act = tok && oc < a.Cout;
        const float r0 = swap1(odd ? on[g][0] : on[g][2]), r1 = swap1(odd ? on[g][1] : on[g][3]);
        f32x4 v = odd ? f32x4{r0, r1, on[g][2], on[g][3]} : f32x4{on[g][0], on[g][1], r0, r1};
        const size_t idx = pixw + (size_t)oc * HW;
        if constexpr (MASK) {
          f32x4 ax = f32x4{1.f, 1.f, 1.f, 1.f};
          if (act) ax = *reinterpret_cast<const f32x4*>(a.aux + idx);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] *= mg_lrelu_mask(ax[e], a.slope);
        }
        if constexpr (BLEND) {  // fade-in: alpha * new branch + (1 - alpha) * old branch (axpby_k's arithmetic)
          f32x4 o4 = f32x4{0.f, 0.f, 0.f, 0.f};
          if (act) o4 = *reinterpret_cast<const f32x4*>(a.other + idx);
          v = bca * v + bcb * o4;
        }
        float pooled = 0.f;
        if constexpr (POOL) {  // the pair's two pooled pixels: each lane holds one image row of both
          const float pe = v[0] + v[1], po = v[2] + v[3];
          pooled = ((odd ? po : pe) + swap1(odd ? pe : po)) * 0.25f;
        }
        if (act) {
          if constexpr (HASY) *reinterpret_cast<f32x4*>(a.y + idx) = v;
          if constexpr (PN) *reinterpret_cast<f32x4*>(a.p + idx) = v * rnw;
          if constexpr (POOL) a.p[pp0 + (size_t)oc * (Ht * Wt)] = pooled;
        }
      }
      return;
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int oc = oc0 + ni * 16 + g;
      if (tok && oc < a.Cout) {
        const size_t idx0 = pix0 + (size_t)oc * HW;
        float2 v0 = make_float2(on[g][0], on[g][1]), v1 = make_float2(on[g][2], on[g][3]);
        if constexpr (MASK) {
          const float2 a0 = *reinterpret_cast<const float2*>(a.aux + idx0);
          const float2 a1 = *reinterpret_cast<const float2*>(a.aux + idx0 + a.W);
          v0.x *= mg_lrelu_mask(a0.x, a.slope);
          v0.y *= mg_lrelu_mask(a0.y, a.slope);
          v1.x *= mg_lrelu_mask(a1.x, a.slope);
          v1.y *= mg_lrelu_mask(a1.y, a.slope);
        }
        if constexpr (BLEND) {
          const float2 o0 = *reinterpret_cast<const float2*>(a.other + idx0);
          const float2 o1 = *reinterpret_cast<const float2*>(a.other + idx0 + a.W);
          v0 = make_float2(bca * v0.x + bcb * o0.x, bca * v0.y + bcb * o0.y);
          v1 = make_float2(bca * v1.x + bcb * o1.x, bca * v1.y + bcb * o1.y);
        }
        if constexpr (HASY) {
          *reinterpret_cast<float2*>(a.y + idx0) = v0;
          *reinterpret_cast<float2*>(a.y + idx0 + a.W) = v1;
        }
        if constexpr (PN) {
          *reinterpret_cast<float2*>(a.p + idx0) = make_float2(v0.x * rnv[0], v0.y * rnv[1]);
          *reinterpret_cast<float2*>(a.p + idx0 + a.W) = make_float2(v1.x * rnv[2], v1.y * rnv[3]);
        }
        if constexpr (POOL) a.p[pp0 + (size_t)oc * (Ht * Wt)] = ((v0.x + v0.y) + (v1.x + v1.y)) * 0.25f;
      }
    }
  };
  // no PixelNorm: one out-channel tile at a time (16 live outputs)
  auto tail = [&](auto mask_, auto pool_, auto hasy_, auto mout_, auto blend_) __attribute__((always_inline)) {
    const float one[4] = {1.f, 1.f, 1.f, 1.f};
    constexpr bool MASKED = decltype(mask_)::value != 0;
#pragma unroll
    for (int ni = 0; ni < NIW; ++ni) {
      float on[4][4];
      if (MASKED && a.bias == nullptr) transform(ni, on, std::false_type{});  // MASK_AUX excludes LRELU; no bias: plain A^T M A
      else transform(ni, on, std::true_type{});
      store_tile(ni, on, one, mask_, std::false_type{}, pool_, hasy_, mout_, blend_);
    }
  };
  // backward of the fade-in blend and of the two LeakyReLUs in front of it, on the data-gradient conv that produces the blend's
  // gradient (blend_lrelu_bwd_k's arithmetic): y = (alpha * acc) * lrelu'(new branch, tile mask a.mi),
  // p = ((1 - alpha) * acc) * lrelu'(old branch activation a.other)
  auto tail_blend_bwd = [&]() __attribute__((always_inline)) {
    const float ca = a.coef[0], cb = a.coef[1];
#pragma unroll
    for (int ni = 0; ni < NIW; ++ni) {
      float on[4][4];
      transform(ni, on, std::false_type{});
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int oc = oc0 + ni * 16 + g;
        const bool act = tok && oc < a.Cout;
        const unsigned mb = act ? (unsigned)a.mi[pp0 + (size_t)oc * (Ht * Wt)] : 15u;
        float oa[4], ob[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          oa[q] = (ca * on[g][q]) * (((mb >> q) & 1u) ? 1.f : a.slope);
          ob[q] = cb * on[g][q];
        }
        if (wide) {
          const float a0 = swap1(odd ? oa[0] : oa[2]), a1 = swap1(odd ? oa[1] : oa[3]);
          const float b0 = swap1(odd ? ob[0] : ob[2]), b1 = swap1(odd ? ob[1] : ob[3]);
          const f32x4 va = odd ? f32x4{a0, a1, oa[2], oa[3]} : f32x4{oa[0], oa[1], a0, a1};
          f32x4 vb = odd ? f32x4{b0, b1, ob[2], ob[3]} : f32x4{ob[0], ob[1], b0, b1};
          if (act) {
            const size_t idx = pixw + (size_t)oc * HW;
            const f32x4 ax = *reinterpret_cast<const f32x4*>(a.other + idx);
#pragma unroll
            for (int e = 0; e < 4; ++e) vb[e] *= mg_lrelu_mask(ax[e], a.slope);
            *reinterpret_cast<f32x4*>(a.y + idx) = va;
            *reinterpret_cast<f32x4*>(a.p + idx) = vb;
          }
        } else if (act) {
          const size_t idx0 = pix0 + (size_t)oc * HW;
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const float2 ax = *reinterpret_cast<const float2*>(a.other + idx0 + (size_t)i * a.W);
            *reinterpret_cast<float2*>(a.y + idx0 + (size_t)i * a.W) = make_float2(oa[2 * i], oa[2 * i + 1]);
            *reinterpret_cast<float2*>(a.p + idx0 + (size_t)i * a.W) =
                make_float2(ob[2 * i] * mg_lrelu_mask(ax.x, a.slope), ob[2 * i + 1] * mg_lrelu_mask(ax.y, a.slope));
          }
        }
      }
    }
  };
  // MG_CONV_UNPOOL: y (N,Cout,2H,2W) = AvgPool2d backward of the result, times the LeakyReLU mask of the layer below it
  // (aux: one byte per OUTPUT PIXEL of this convolution = per 2x2 block of y).  A lane of the wide path holds one image row of
  // a tile pair (4 pixels, 4 mask bytes = one dword) and writes 2 rows x 8 floats; y index of pixel (Y, X) = 4*idx - 2*X.
  auto tail_unpool = [&]() __attribute__((always_inline)) {
    const unsigned char* mb = reinterpret_cast<const unsigned char*>(a.aux);
    const float qh = 0.25f, ql = 0.25f * a.slope;
    const int W2 = 2 * a.W;
#pragma unroll
    for (int ni = 0; ni < NIW; ++ni) {
      float on[4][4];
      transform(ni, on, std::false_type{});
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int oc = oc0 + ni * 16 + g;
        const bool act = tok && oc < a.Cout;
        if (wide) {
          const float r0 = swap1(odd ? on[g][0] : on[g][2]), r1 = swap1(odd ? on[g][1] : on[g][3]);
          const f32x4 v = odd ? f32x4{r0, r1, on[g][2], on[g][3]} : f32x4{on[g][0], on[g][1], r0, r1};
          if (act) {
            const size_t idx = pixw + (size_t)oc * HW;
            const unsigned mw = *reinterpret_cast<const unsigned*>(mb + idx);
            float* dst = a.y + 4 * idx - 2 * (size_t)(2 * (TX - (odd ? 1 : 0)));
#pragma unroll
            for (int r = 0; r < 2; ++r) {
              f32x4 o[2];
#pragma unroll
              for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int j = 0; j < 2; ++j) o[e >> 1][(e & 1) * 2 + j] = v[e] * (((mw >> (8 * e + 2 * r + j)) & 1u) ? qh : ql);
              *reinterpret_cast<f32x4*>(dst + (size_t)r * W2) = o[0];
              *reinterpret_cast<f32x4*>(dst + (size_t)r * W2 + 4) = o[1];
            }
          }
        } else if (act) {
          const size_t idx0 = pix0 + (size_t)oc * HW;
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const unsigned mw = *reinterpret_cast<const unsigned short*>(mb + idx0 + (size_t)i * a.W);  // pixels (i, 0), (i, 1)
            float* dst = a.y + 4 * (idx0 + (size_t)i * a.W) - 2 * (size_t)(2 * TX);
#pragma unroll
            for (int r = 0; r < 2; ++r) {
              f32x4 o;
#pragma unroll
              for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int j = 0; j < 2; ++j) o[e * 2 + j] = on[g][2 * i + e] * (((mw >> (8 * e + 2 * r + j)) & 1u) ? qh : ql);
              *reinterpret_cast<f32x4*>(dst + (size_t)r * W2) = o;
            }
          }
        }
      }
    }
  };
  // PixelNorm: all channels of the pixel first (sum of squares over ni, g in-lane, rq by shuffles, wave groups via LDS)
  auto tail_pn = [&](auto hasy_) __attribute__((always_inline)) {
    float o[NIW][4][4];
#pragma unroll
    for (int ni = 0; ni < NIW; ++ni) transform(ni, o[ni], std::true_type{});
    float rnv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {  // padded out-channels hold exact zeros (zero filters, no bias) and add nothing
      float t = 0.f;
#pragma unroll
      for (int ni = 0; ni < NIW; ++ni)
#pragma unroll
        for (int g = 0; g < 4; ++g) t += o[ni][g][q] * o[ni][g][q];
      t += __shfl_xor(t, 16);
      t += __shfl_xor(t, 32);
      rnv[q] = t;
      if (WC > 1 && rq == 0) red[(wc * TPB + tl) * 4 + q] = t;
    }
    if (WC > 1) {
      __syncthreads();
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float t = 0.f;
#pragma unroll
        for (int w2 = 0; w2 < WC; ++w2) t += red[(w2 * TPB + tl) * 4 + q];
        rnv[q] = t;
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) rnv[q] = 1.0f / sqrtf(rnv[q] / (float)a.Cout + PN_EPS);
#pragma unroll
    for (int ni = 0; ni < NIW; ++ni)
      store_tile(ni, o[ni], rnv, std::integral_constant<int, 0>{}, std::true_type{}, std::false_type{}, hasy_, std::false_type{},
                 std::false_type{});
    if (tok && rq == 0 && wc == 0 && a.rn != nullptr && blockIdx.y == 0) {
      const size_t r0 = ((size_t)n * a.H + 2 * TY) * a.W + 2 * TX;
      *reinterpret_cast<float2*>(a.rn + r0) = make_float2(rnv[0], rnv[1]);
      *reinterpret_cast<float2*>(a.rn + r0 + a.W) = make_float2(rnv[2], rnv[3]);
    }
  };
  using T_ = std::true_type;
  using F_ = std::false_type;
  using I0_ = std::integral_constant<int, 0>;
  using I1_ = std::integral_constant<int, 1>;
  using I2_ = std::integral_constant<int, 2>;
  if (a.flags & MG_CONV_PIXNORM) {
    if (a.y != nullptr) tail_pn(T_{});
    else tail_pn(F_{});
  } else if (a.flags & MG_CONV_UNPOOL) {
    tail_unpool();
  } else if (a.flags & WF_BLEND_BWD) {
    tail_blend_bwd();
  } else if (a.flags & WF_BLEND) {
    if (a.flags & MG_CONV_MASK_BYTES) tail(I2_{}, F_{}, T_{}, F_{}, T_{});  // tangent pass through the blend
    else tail(I0_{}, F_{}, T_{}, T_{}, T_{});                                // forward: blend + tile mask of the new branch
  } else if (a.flags & MG_CONV_MASK_AUX) {
    if (a.flags & MG_CONV_MASK_BYTES) tail(I2_{}, T_{}, F_{}, F_{}, F_{});  // pooled result only
    else if (a.flags & MG_CONV_POOL_OUT) tail(I1_{}, T_{}, T_{}, F_{}, F_{});
    else tail(I1_{}, F_{}, T_{}, F_{}, F_{});
  } else {
    if (a.flags & MG_CONV_MASK_OUT) tail(I0_{}, T_{}, F_{}, T_{}, F_{});  // pooled result + tile mask bytes instead of y
    else if (a.flags & MG_CONV_POOL_OUT) tail(I0_{}, T_{}, T_{}, F_{}, F_{});
    else tail(I0_{}, F_{}, T_{}, F_{}, F_{});
  }

// vssr_api.hip — extern "C" entry points declared in include/vssr_eval.h.
#include <cmath>
#include <cstdarg>

#include "vssr_internal.h"

namespace vssr {

const char *const kKernelClassNames[KC_COUNT] = {
    "neighbor_list", "embed", "message_mlp", "edge_message_fwd", "update_fwd", "readout",
    "update_bwd", "edge_message_bwd", "message_mlp_bwd", "finalize", "tersoff", "layer0_factorised_fwd",
    "layer0_factorised_bwd"};

static thread_local std::string g_create_error;   // per thread: handles may be created concurrently (one host thread per engine)

int set_err(vssr_handle *h, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (h) h->err = buf;
    else g_create_error = buf;
    return code;
}

// ---- profiler ------------------------------------------------------------------------------------------
hipEvent_t Profiler::get_event() {
    if (!pool.empty()) {
        hipEvent_t e = pool.back();
        pool.pop_back();
        return e;
    }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}
void Profiler::begin(int kc, hipStream_t s) {
    if (!enabled) return;
    Rec r{kc, get_event(), get_event()};
    (void)hipEventRecord(r.a, s);
    pending.push_back(r);
}
void Profiler::end(hipStream_t s) {
    if (!enabled) return;
    (void)hipEventRecord(pending.back().b, s);
}
void Profiler::collect() {
    for (auto &r : pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            total_ms[r.kc] += ms;
            launches[r.kc] += 1;
        }
        pool.push_back(r.a);
        pool.push_back(r.b);
    }
    pending.clear();
}
void Profiler::reset() {
    collect();
    for (int k = 0; k < KC_COUNT; ++k) { launches[k] = 0; total_ms[k] = 0; }
}
void Profiler::destroy() {
    collect();
    for (auto e : pool) (void)hipEventDestroy(e);
    pool.clear();
}

// ---- weights ------------------------------------------------------------------------------------------
static void transpose(const float *src, int rows, int cols, float *dst) {  // dst[c][r] = src[r][c]
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < cols; ++c) dst[(size_t)c * rows + r] = src[(size_t)r * cols + c];
}

static int upload_weights(vssr_handle *h, const vssr_painn_config *cfg) {
    const int M = cfg->n_models, L = cfg->num_conv, R = cfg->n_rbf, H = cfg->readout_hidden, NE = cfg->n_embed;
    const size_t per_layer = (size_t)F * F + F + (size_t)F3 * F + F3 + (size_t)F3 * R + F3 + 2 * (size_t)F * F +
                             (size_t)F * 2 * F + F + (size_t)F3 * F + F3;
    const size_t blob_len = (size_t)NE * F + L * per_layer + (size_t)H * F + H + H + 1;
    if (cfg->weights_len != blob_len)
        return set_err(h, VSSR_E_BADARG, "weights_len %llu does not match the layout (%zu floats)",
                       (unsigned long long)cfg->weights_len, blob_len);
    // device image per model: [blob][transposed copies]
    const size_t t_per_layer = (size_t)F * F + (size_t)F * F3 + 2 * (size_t)F * F + (size_t)2 * F * F + (size_t)F * F3;
    const size_t img_len = blob_len + L * t_per_layer + (size_t)F * H;
    std::vector<float> img(img_len * M);
    std::vector<ModelW> table(M);
    if (h->weights.ensure(img.size() * sizeof(float)))
        return set_err(h, VSSR_E_NOMEM, "weights: out of device memory");
    float *dbase = h->weights.as<float>();
    // fp16-split fragment-order copies of the node-GEMM weights: 22 F^2 elements x 4 B per (model, layer)
    const size_t node16_per_layer = (size_t)22 * F * F;   // dwords
    const size_t node16_readout = (size_t)2 * H * F;      // W5 and W5^T (used when the width is a multiple of 32)
    std::vector<unsigned> node16(node16_per_layer * L * M + node16_readout * M);
    for (int m = 0; m < M; ++m) {
        float *hb = img.data() + (size_t)m * img_len;
        float *db = dbase + (size_t)m * img_len;
        memcpy(hb, cfg->weights[m], blob_len * sizeof(float));
        for (size_t t = 0; t < blob_len; ++t)
            if (!std::isfinite(hb[t])) return set_err(h, VSSR_E_BADARG, "model %d: non-finite weight", m);
        ModelW &W = table[m];
        size_t o = 0, to = blob_len;
        auto take = [&](size_t n) { size_t r = o; o += n; return r; };
        auto taket = [&](size_t n) { size_t r = to; to += n; return r; };
        W.embed = db + take((size_t)NE * F);
        for (int l = 0; l < L; ++l) {
            LayerW &Lw = W.layer[l];
            size_t w1 = take((size_t)F * F), b1 = take(F), w2 = take((size_t)F3 * F), b2 = take(F3);
            size_t wd = take((size_t)F3 * R), bd = take(F3), u = take((size_t)F * F), v = take((size_t)F * F);
            size_t w3 = take((size_t)F * 2 * F), b3 = take(F), w4 = take((size_t)F3 * F), b4 = take(F3);
            size_t w1t = taket((size_t)F * F), w2t = taket((size_t)F * F3), ut = taket((size_t)F * F);
            size_t vt = taket((size_t)F * F), w3t = taket((size_t)2 * F * F), w4t = taket((size_t)F * F3);
            transpose(hb + w1, F, F, hb + w1t);
            transpose(hb + w2, F3, F, hb + w2t);
            transpose(hb + u, F, F, hb + ut);
            transpose(hb + v, F, F, hb + vt);
            transpose(hb + w3, F, 2 * F, hb + w3t);
            transpose(hb + w4, F3, F, hb + w4t);
            Lw.W1 = db + w1; Lw.W1t = db + w1t; Lw.b1 = db + b1;
            Lw.W2 = db + w2; Lw.W2t = db + w2t; Lw.b2 = db + b2;
            Lw.Wd = db + wd; Lw.bd = db + bd;
            Lw.U = db + u; Lw.Ut = db + ut; Lw.V = db + v; Lw.Vt = db + vt;
            Lw.W3 = db + w3; Lw.W3t = db + w3t; Lw.b3 = db + b3;
            Lw.W4 = db + w4; Lw.W4t = db + w4t; Lw.b4 = db + b4;
            // fp16-split MFMA fragment-order copies of the eleven node-GEMM matrices (painn_node_mfma.hip)
            {   // [U;V]^T: rows g (input feature of U/V), K = 2F: k<F -> U[k][g], k>=F -> V[k-F][g]
                std::vector<float> uvt((size_t)F * 2 * F);
                for (int g = 0; g < F; ++g)
                    for (int k = 0; k < F; ++k) {
                        uvt[(size_t)g * 2 * F + k] = hb[u + (size_t)k * F + g];
                        uvt[(size_t)g * 2 * F + F + k] = hb[v + (size_t)k * F + g];
                    }
                // fp16-split copies, same order as the q* pointers are assigned below
                unsigned *q = node16.data() + ((size_t)m * L + l) * node16_per_layer;
                auto put16 = [&](const float *src, int rows, int K) {
                    pack_mfma_tiles16(src, rows, K, q);
                    q += (size_t)rows * K;
                };
                put16(hb + w1, F, F); put16(hb + w2, F3, F); put16(hb + u, F, F); put16(hb + v, F, F);
                put16(hb + w3, F, 2 * F); put16(hb + w4, F3, F); put16(hb + w1t, F, F); put16(hb + w2t, F, F3);
                put16(hb + w4t, F, F3); put16(hb + w3t, 2 * F, F); put16(uvt.data(), F, 2 * F);
            }
        }
        size_t w5 = take((size_t)H * F), b5 = take(H), w6 = take(H), b6 = take(1);
        size_t w5t = taket((size_t)F * H);
        transpose(hb + w5, H, F, hb + w5t);
        W.W5 = db + w5; W.W5t = db + w5t; W.b5 = db + b5; W.w6 = db + w6; W.b6 = db + b6;
        if (H % 32 == 0) {   // fragment-order pieces of the readout matrices (matrix-pipe readout, painn_node_mfma.hip)
            unsigned *q = node16.data() + node16_per_layer * L * M + node16_readout * m;
            pack_mfma_tiles16(hb + w5, H, F, q);
            pack_mfma_tiles16(hb + w5t, F, H, q + (size_t)H * F);
        }
    }
    VSSR_HIP(h, hipMemcpy(dbase, img.data(), img.size() * sizeof(float), hipMemcpyHostToDevice));
    {   // node-GEMM weights as fp16 pieces (painn_node_mfma.hip)
        if (h->node16.ensure(node16.size() * sizeof(unsigned))) return set_err(h, VSSR_E_NOMEM, "split node weights");
        VSSR_HIP(h, hipMemcpy(h->node16.p, node16.data(), node16.size() * sizeof(unsigned), hipMemcpyHostToDevice));
        for (int m = 0; m < M; ++m)
            for (int l = 0; l < L; ++l) {
                LayerW &Lw = table[m].layer[l];
                const uint4 *q = h->node16.as<uint4>() + ((size_t)m * L + l) * node16_per_layer / 4;
                auto next = [&](int rows, int K) { const uint4 *r = q; q += (size_t)rows * K / 4; return r; };
                Lw.qW1 = next(F, F); Lw.qW2 = next(F3, F); Lw.qU = next(F, F); Lw.qV = next(F, F);
                Lw.qW3 = next(F, 2 * F); Lw.qW4 = next(F3, F); Lw.qW1t = next(F, F); Lw.qW2t = next(F, F3);
                Lw.qW4t = next(F, F3); Lw.qW3t = next(2 * F, F); Lw.qUVt = next(F, 2 * F);
            }
        for (int m = 0; m < M; ++m) {
            const uint4 *q = h->node16.as<uint4>() + (node16_per_layer * L * M + node16_readout * m) / 4;
            table[m].qW5 = q;
            table[m].qW5t = q + (size_t)H * F / 4;
        }
    }
    {   // radial-filter weights split into fp16 pieces in MFMA operand order (painn_edge_mfma.hip), per model / layer
        const size_t per_layer16 = (size_t)F3 * 4 * 8;   // dwords
        std::vector<unsigned> w16(per_layer16 * L * M);
        for (int m = 0; m < M; ++m) {
            const float *hb = img.data() + (size_t)m * img_len;
            size_t o = (size_t)NE * F;
            for (int l = 0; l < L; ++l) {
                const float *Wd = hb + o + (size_t)F * F + F + (size_t)F3 * F + F3;
                build_wd16(Wd, Wd + (size_t)F3 * 20, w16.data() + ((size_t)m * L + l) * per_layer16);
                o += per_layer;
            }
        }
        if (h->wd16.ensure(w16.size() * sizeof(unsigned))) return set_err(h, VSSR_E_NOMEM, "split filter weights");
        VSSR_HIP(h, hipMemcpy(h->wd16.p, w16.data(), w16.size() * sizeof(unsigned), hipMemcpyHostToDevice));
        for (int m = 0; m < M; ++m)
            for (int l = 0; l < L; ++l)
                table[m].layer[l].wd16 = h->wd16.as<uint4>() + ((size_t)m * L + l) * per_layer16 / 4;
    }
    {   // layer-0 species factorisation tables (painn_l0.hip), from layer 0 of every model
        const size_t per_model = (size_t)NE * 2 * 24 * F;
        std::vector<float> A(per_model * M), At(per_model * M);
        for (int m = 0; m < M; ++m) {
            const float *hb = img.data() + (size_t)m * img_len;
            size_t o = (size_t)NE * F;
            const float *W1 = hb + o; o += (size_t)F * F;
            const float *b1 = hb + o; o += F;
            const float *W2 = hb + o; o += (size_t)F3 * F;
            const float *b2 = hb + o; o += F3;
            const float *Wd = hb + o; o += (size_t)F3 * R;
            const float *bd = hb + o;
            l0_build_tables(hb, W1, b1, W2, b2, Wd, bd, NE, A.data() + per_model * m, At.data() + per_model * m);
        }
        // the kernels read the fp16-split fragment-order copies (painn_l0.hip); the fp32 tables stay on the host
        const size_t pk = l0_packed_dwords(NE);
        std::vector<unsigned> A16(pk * M), At16(pk * M);
        for (int m = 0; m < M; ++m) l0_pack_tables(A.data() + per_model * m, NE, A16.data() + pk * m, At16.data() + pk * m);
        if (h->d_l0A.ensure(A16.size() * sizeof(unsigned)) || h->d_l0At.ensure(At16.size() * sizeof(unsigned)))
            return set_err(h, VSSR_E_NOMEM, "layer-0 tables");
        VSSR_HIP(h, hipMemcpy(h->d_l0A.p, A16.data(), A16.size() * sizeof(unsigned), hipMemcpyHostToDevice));
        VSSR_HIP(h, hipMemcpy(h->d_l0At.p, At16.data(), At16.size() * sizeof(unsigned), hipMemcpyHostToDevice));
    }
    if (h->model_table.ensure(sizeof(ModelW) * M)) return set_err(h, VSSR_E_NOMEM, "model table");
    VSSR_HIP(h, hipMemcpy(h->model_table.p, table.data(), sizeof(ModelW) * M, hipMemcpyHostToDevice));
    return VSSR_OK;
}

static int common_init(vssr_handle *h, int device) {
    h->device = device;
    VSSR_HIP(h, hipSetDevice(device));
    VSSR_HIP(h, hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    VSSR_HIP(h, hipHostMalloc((void **)&h->h_counters, sizeof(int) * 4));
    h->h_counters[0] = h->h_counters[1] = h->h_counters[2] = 0;
    return VSSR_OK;
}

static void cell_host_setup(const double *cell, const uint8_t *pbc, double cutoff, double inv[9], int nimg[3],
                            bool &ok) {
    const double *a = cell, *b = cell + 3, *c = cell + 6;
    double bc[3] = {b[1] * c[2] - b[2] * c[1], b[2] * c[0] - b[0] * c[2], b[0] * c[1] - b[1] * c[0]};
    double ca[3] = {c[1] * a[2] - c[2] * a[1], c[2] * a[0] - c[0] * a[2], c[0] * a[1] - c[1] * a[0]};
    double ab[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
    double vol = a[0] * bc[0] + a[1] * bc[1] + a[2] * bc[2];
    ok = true;
    for (int x = 0; x < 9; ++x) inv[x] = 0.0;
    nimg[0] = nimg[1] = nimg[2] = 0;
    if (!(pbc[0] || pbc[1] || pbc[2])) return;
    if (std::fabs(vol) < 1e-12) { ok = false; return; }
    for (int x = 0; x < 3; ++x) { inv[x] = bc[x] / vol; inv[3 + x] = ca[x] / vol; inv[6 + x] = ab[x] / vol; }
    double hgt[3] = {std::fabs(vol) / std::sqrt(bc[0] * bc[0] + bc[1] * bc[1] + bc[2] * bc[2]),
                     std::fabs(vol) / std::sqrt(ca[0] * ca[0] + ca[1] * ca[1] + ca[2] * ca[2]),
                     std::fabs(vol) / std::sqrt(ab[0] * ab[0] + ab[1] * ab[1] + ab[2] * ab[2])};
    for (int k = 0; k < 3; ++k)
        if (pbc[k]) nimg[k] = (int)std::floor(cutoff / hgt[k]) + 1;
}

static double handle_cutoff(const vssr_handle *h) {
    return h->kind == 2 ? h->ters_cutmax : h->kind == 3 ? h->eam_grid.cutoff : (double)h->cutoff;
}

static int run_any(vssr_handle *h, uint32_t want) {
    h->last_want = want;
    return h->kind == 2 ? tersoff_run(h, want) : h->kind == 3 ? eam_run(h, want) : painn_run(h, want);
}

// synchronise; if the neighbor capacity overflowed, grow and rerun
static int sync_and_check(vssr_handle *h) {
    const uint32_t want = h->last_want;   // a rerun after a capacity overflow produces what the original run was asked for
    for (int attempt = 0; attempt < 4; ++attempt) {
        VSSR_HIP(h, hipStreamSynchronize(h->stream));
        h->prof.collect();
        if (!h->ran || !h->h_counters[2]) return VSSR_OK;
        if (h->h_counters[0] <= 0)
            return set_err(h, VSSR_E_CAPACITY, "neighbor list exceeds 2^31 slots");
        h->slot_cap = (int64_t)h->h_counters[0] + (h->cap_tight ? 0 : (int64_t)h->h_counters[0] / 8) + 64;
        int rc = run_any(h, want);
        if (rc) return rc;
    }
    return set_err(h, VSSR_E_CAPACITY, "neighbor list capacity could not be satisfied");
}

}  // namespace vssr

using namespace vssr;

extern "C" {

int vssr_abi_version(void) { return VSSR_ABI_VERSION; }

const char *vssr_last_error(const vssr_handle *h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int vssr_create(const vssr_painn_config *cfg, vssr_handle **out) {
    if (!cfg || !out) return set_err(nullptr, VSSR_E_BADARG, "null argument");
    *out = nullptr;
    if (cfg->struct_size != sizeof(vssr_painn_config))
        return set_err(nullptr, VSSR_E_BADARG, "vssr_painn_config size mismatch (%u vs %zu)", cfg->struct_size,
                       sizeof(vssr_painn_config));
    if (cfg->feat_dim != F || cfg->n_rbf != 20)
        return set_err(nullptr, VSSR_E_BADARG, "only feat_dim=128, n_rbf=20 are compiled (got %d, %d)", cfg->feat_dim,
                       cfg->n_rbf);
    if (cfg->n_models < 1 || cfg->n_models > MAX_MODELS || cfg->num_conv < 1 || cfg->num_conv > MAX_LAYERS ||
        cfg->readout_hidden < 1 || cfg->readout_hidden > F || cfg->n_embed < 1 || !cfg->weights ||
        !(cfg->cutoff > 0) || !(cfg->model_units_per_ev > 0))
        return set_err(nullptr, VSSR_E_BADARG, "bad PaiNN configuration");
    if (cfg->excl_vol && (cfg->excl_power < 1 || cfg->excl_power > 64 || !(cfg->excl_sigma > 0)))
        return set_err(nullptr, VSSR_E_BADARG, "excluded volume: power %d (1 .. 64) / sigma %g out of range", cfg->excl_power,
                       (double)cfg->excl_sigma);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return set_err(nullptr, VSSR_E_DEVICE, "no HIP device available (this backend has no CPU fallback)");
    if (cfg->device < 0 || cfg->device >= ndev) return set_err(nullptr, VSSR_E_BADARG, "device %d out of range", cfg->device);
    vssr_handle *h = new vssr_handle();
    h->kind = 1;
    h->n_models = cfg->n_models; h->n_rbf = cfg->n_rbf; h->num_conv = cfg->num_conv; h->n_embed = cfg->n_embed;
    h->readout_hidden = cfg->readout_hidden; h->cutoff = cfg->cutoff; h->excl_vol = cfg->excl_vol;
    h->excl_power = cfg->excl_power; h->excl_sigma = cfg->excl_sigma; h->units_per_ev = cfg->model_units_per_ev;
    int rc = common_init(h, cfg->device);
    if (!rc) rc = upload_weights(h, cfg);
    if (!rc) rc = node_mfma_init(h);
    if (!rc) rc = edge_mfma_init(h);
    if (!rc) rc = l0_mfma_init(h);
    if (const char *e = getenv("VSSR_EDGE_IMPL")) h->edge_impl = (strcmp(e, "gather") == 0) ? 0 : 1;
    if (const char *e = getenv("VSSR_L0_FACTORISE")) h->l0_enabled = atoi(e);
    if (const char *e = getenv("VSSR_UPD_SAVE")) h->upd_save = atoi(e);
    if (const char *e = getenv("VSSR_DEBUG_KEEP")) h->debug_keep = atoi(e);
    // test knobs: send chains above these atom counts to the next class (8-feature slices / gather kernels) although they fit
    if (const char *e = getenv("VSSR_EDGE_FS16_MAX")) h->fs16_max_atoms = atoi(e);
    if (const char *e = getenv("VSSR_EDGE_FS8_MAX")) h->fs8_max_atoms = atoi(e);
    if (const char *e = getenv("VSSR_EDGE_BWD_MPASS")) h->bwd_multi_pass = atoi(e);
    if (const char *e = getenv("VSSR_EDGE_SUB_CHUNK")) { const int c = atoi(e); if (c >= 8) { h->sub_chunk_fwd = c; h->sub_chunk_bwd = c; } }
    if (const char *e = getenv("VSSR_EDGE_FWD_2PASS")) { const int w = atoi(e); h->fwd_two_pass = (w == 8 || w == 16) ? w : w ? 16 : 0; }
    if (!rc && cfg->offset_per_z) {
        h->has_offset = true;
        h->offset_const = cfg->offset_const;
        if (h->offset_per_z.ensure(sizeof(double) * cfg->n_embed)) rc = set_err(h, VSSR_E_NOMEM, "offset table");
        else if (hipMemcpy(h->offset_per_z.p, cfg->offset_per_z, sizeof(double) * cfg->n_embed,
                           hipMemcpyHostToDevice) != hipSuccess)
            rc = set_err(h, VSSR_E_DEVICE, "offset table upload failed");
    }
    if (rc) {
        g_create_error = h->err;
        vssr_destroy(h);
        return rc;
    }
    *out = h;
    return VSSR_OK;
}

int vssr_tersoff_create(int32_t device, int32_t n_types, const double *params, vssr_handle **out) {
    if (!params || !out || n_types < 1 || n_types > 8) return set_err(nullptr, VSSR_E_BADARG, "bad tersoff arguments");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return set_err(nullptr, VSSR_E_DEVICE, "no HIP device available (this backend has no CPU fallback)");
    if (device < 0 || device >= ndev) return set_err(nullptr, VSSR_E_BADARG, "device %d out of range", device);
    vssr_handle *h = new vssr_handle();
    h->kind = 2;
    h->n_types = n_types;
    h->n_embed = n_types;
    int rc = common_init(h, device);
    const size_t np = (size_t)n_types * n_types * n_types;
    double cutmax = 0;
    for (size_t t = 0; t < np && !rc; ++t) {
        const double *p = params + 14 * t;
        if (!(p[0] == 1.0 || p[0] == 3.0) || !(p[11] > 0) || !(p[4] != 0)) rc = set_err(h, VSSR_E_BADARG, "bad tersoff entry %zu", t);
        if (p[10] + p[11] > cutmax) cutmax = p[10] + p[11];
    }
    h->ters_cutmax = cutmax;
    if (!rc && h->ters_params.ensure(sizeof(double) * 14 * np)) rc = set_err(h, VSSR_E_NOMEM, "tersoff params");
    if (!rc && hipMemcpy(h->ters_params.p, params, sizeof(double) * 14 * np, hipMemcpyHostToDevice) != hipSuccess)
        rc = set_err(h, VSSR_E_DEVICE, "tersoff params upload failed");
    if (rc) {
        g_create_error = h->err;
        vssr_destroy(h);
        return rc;
    }
    *out = h;
    return VSSR_OK;
}

int vssr_tersoff_create_from_text(int32_t device, const char *param_text, int32_t n_species, const char *const *species,
                                  vssr_handle **out) {
    if (!param_text || !species || !out || n_species < 1 || n_species > 8)
        return set_err(nullptr, VSSR_E_BADARG, "bad tersoff arguments");
    *out = nullptr;
    std::vector<std::string> tok;
    {
        std::string line, text(param_text);
        size_t pos = 0;
        while (pos <= text.size()) {
            size_t nl = text.find('\n', pos);
            if (nl == std::string::npos) nl = text.size();
            line = text.substr(pos, nl - pos);
            pos = nl + 1;
            const size_t hash = line.find('#');
            if (hash != std::string::npos) line.erase(hash);
            size_t i = 0;
            while (i < line.size()) {
                while (i < line.size() && isspace((unsigned char)line[i])) ++i;
                size_t j = i;
                while (j < line.size() && !isspace((unsigned char)line[j])) ++j;
                if (j > i) tok.push_back(line.substr(i, j - i));
                i = j;
            }
        }
    }
    if (tok.empty() || tok.size() % 17) return set_err(nullptr, VSSR_E_BADARG, "tersoff file: token count is not a multiple of 17");
    auto index_of = [&](const std::string &s) {
        for (int t = 0; t < n_species; ++t)
            if (species[t] && s == species[t]) return t;
        return -1;
    };
    const size_t np = (size_t)n_species * n_species * n_species;
    std::vector<double> params(14 * np, 0.0);
    std::vector<char> seen(np, 0);
    for (size_t o = 0; o < tok.size(); o += 17) {
        const int a = index_of(tok[o]), b = index_of(tok[o + 1]), c = index_of(tok[o + 2]);
        if (a < 0 || b < 0 || c < 0) continue;   // entry of another element
        const size_t e = ((size_t)a * n_species + b) * n_species + c;
        for (int k = 0; k < 14; ++k) {
            char *end = nullptr;
            params[14 * e + k] = strtod(tok[o + 3 + k].c_str(), &end);
            if (!end || *end) return set_err(nullptr, VSSR_E_BADARG, "tersoff file: bad number '%s'", tok[o + 3 + k].c_str());
        }
        seen[e] = 1;
    }
    for (size_t e = 0; e < np; ++e)
        if (!seen[e]) return set_err(nullptr, VSSR_E_BADARG, "tersoff file lacks entries for some species triplets");
    return vssr_tersoff_create(device, n_species, params.data(), out);
}

int vssr_eam_create(int32_t device, const vssr_eam_grid *grid, const double *frho, const double *zr, const double *rhor,
                    vssr_handle **out) {
    if (!grid || !frho || !zr || !rhor || !out) return set_err(nullptr, VSSR_E_BADARG, "null EAM argument");
    *out = nullptr;
    if (grid->nrho < 5 || grid->nr < 5 || !(grid->drho > 0) || !(grid->dr > 0) || !(grid->cutoff > 0))
        return set_err(nullptr, VSSR_E_BADARG, "bad EAM grid");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return set_err(nullptr, VSSR_E_DEVICE, "no HIP device available (this backend has no CPU fallback)");
    if (device < 0 || device >= ndev) return set_err(nullptr, VSSR_E_BADARG, "device %d out of range", device);
    for (int k = 0; k < grid->nrho; ++k)
        if (!std::isfinite(frho[k])) return set_err(nullptr, VSSR_E_BADARG, "non-finite EAM table entry");
    for (int k = 0; k < grid->nr; ++k)
        if (!std::isfinite(zr[k]) || !std::isfinite(rhor[k])) return set_err(nullptr, VSSR_E_BADARG, "non-finite EAM table entry");
    vssr_handle *h = new vssr_handle();
    h->kind = 3;
    h->n_types = 1;
    h->n_embed = 1;
    h->eam_grid = *grid;
    int rc = common_init(h, device);
    // spline tables: frho [nrho + 1][7] | rhor [nr + 1][7] | z2r [nr + 1][7]   (LAMMPS file2array + array2spline for one file)
    const size_t nF = 7 * (size_t)(grid->nrho + 1), nR = 7 * (size_t)(grid->nr + 1);
    std::vector<double> tab(nF + 2 * nR), z2r(grid->nr);
    for (int k = 0; k < grid->nr; ++k) z2r[k] = 27.2 * 0.529 * zr[k] * zr[k];
    eam_build_spline(frho, grid->nrho, grid->drho, tab.data());
    eam_build_spline(rhor, grid->nr, grid->dr, tab.data() + nF);
    eam_build_spline(z2r.data(), grid->nr, grid->dr, tab.data() + nF + nR);
    if (!rc && h->ters_params.ensure(sizeof(double) * tab.size())) rc = set_err(h, VSSR_E_NOMEM, "EAM tables");
    if (!rc && hipMemcpy(h->ters_params.p, tab.data(), sizeof(double) * tab.size(), hipMemcpyHostToDevice) != hipSuccess)
        rc = set_err(h, VSSR_E_DEVICE, "EAM table upload failed");
    if (rc) {
        g_create_error = h->err;
        vssr_destroy(h);
        return rc;
    }
    *out = h;
    return VSSR_OK;
}

void vssr_destroy(vssr_handle *h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    h->prof.destroy();
    DevBuf *bufs[] = {&h->weights, &h->model_table, &h->offset_per_z, &h->ters_params, &h->d_pos, &h->d_wpos,
                      &h->d_wrap, &h->d_Z, &h->d_atom_cfg, &h->d_cfg_start, &h->d_cell, &h->d_invcell, &h->d_nimg,
                      &h->d_pbc, &h->d_deg, &h->d_row_start, &h->d_edge, &h->d_edge_S, &h->d_rev, &h->d_counters, &h->d_tile_sums, &h->d_erec, &h->d_rho, &h->d_dist, &h->d_rho16, &h->d_drho16, &h->d_zslot, &h->d_bundle, &h->d_excl, &h->d_hits, &h->wd16, &h->node16, &h->d_l0A, &h->d_l0At, &h->d_zmap, &h->d_zlist, &h->d_l0T, &h->d_l0Q, &h->d_vel, &h->d_fire, &h->d_fixed, &h->d_relax_steps, &h->d_relax_conv, &h->d_active, &h->d_bfgs_q, &h->d_bfgs_b,
                      &h->d_state, &h->d_gbar, &h->d_energy, &h->d_energy_std, &h->d_energy_models, &h->d_forces,
                      &h->d_forces_std, &h->d_e_atoms, &h->d_ters_e, &h->d_ters_ea, &h->d_ters_f, &h->d_sat, &h->d_sat_out, &h->d_stress, &h->d_traj_pos, &h->d_traj_f, &h->d_traj_e, &h->d_traj_n, &h->d_chain_class, &h->d_class_list, &h->d_upd_save, &h->d_gpart, &h->d_energy64, &h->d_cmp, &h->d_cm, &h->d_bundle_sub, &h->d_bundle_subb};
    for (DevBuf *b : bufs) b->release();
    if (h->h_counters) (void)hipHostFree(h->h_counters);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

int vssr_batch_upload(vssr_handle *h, int32_t n_cfg, const int32_t *n_atoms, const int32_t *Z, const double *pos,
                      const double *cell, const uint8_t *pbc) {
    if (!h) return VSSR_E_BADARG;
    if (n_cfg < 1 || !n_atoms || !Z || !pos || !cell || !pbc) return set_err(h, VSSR_E_BADARG, "null or empty batch");
    VSSR_HIP(h, hipSetDevice(h->device));
    VSSR_HIP(h, hipStreamSynchronize(h->stream));
    h->batch_valid = false;
    h->ran = false;
    std::vector<int> start(n_cfg + 1, 0);
    for (int b = 0; b < n_cfg; ++b) {
        if (n_atoms[b] < 1) return set_err(h, VSSR_E_BADARG, "configuration %d has %d atoms", b, n_atoms[b]);
        if ((int64_t)start[b] + n_atoms[b] > 2000000000LL) return set_err(h, VSSR_E_BADARG, "batch too large");
        start[b + 1] = start[b] + n_atoms[b];
    }
    const int N = start[n_cfg];
    std::vector<int> atom_cfg(N);
    for (int b = 0; b < n_cfg; ++b)
        for (int i = start[b]; i < start[b + 1]; ++i) atom_cfg[i] = b;
    for (int i = 0; i < N; ++i)
        if (Z[i] < 0 || Z[i] >= h->n_embed)
            return set_err(h, VSSR_E_BADARG, "atom %d: species index %d outside [0,%d)", i, Z[i], h->n_embed);
    for (size_t t = 0; t < (size_t)3 * N; ++t)
        if (!std::isfinite(pos[t])) return set_err(h, VSSR_E_BADARG, "non-finite position");
    std::vector<double> inv((size_t)9 * n_cfg);
    std::vector<int> nimg((size_t)3 * n_cfg);
    const double rc = handle_cutoff(h);
    for (int b = 0; b < n_cfg; ++b) {
        bool ok;
        cell_host_setup(cell + 9 * b, pbc + 3 * b, rc, inv.data() + 9 * b, nimg.data() + 3 * b, ok);
        if (!ok) return set_err(h, VSSR_E_BADARG, "configuration %d: periodic but singular cell", b);
        for (int k = 0; k < 3; ++k)
            if (nimg[3 * b + k] > 100) return set_err(h, VSSR_E_BADARG, "configuration %d: cell too thin for the cutoff", b);
        const long long imgs = (2LL * nimg[3 * b] + 1) * (2 * nimg[3 * b + 1] + 1) * (2 * nimg[3 * b + 2] + 1);
        if (b == 0 || imgs > h->max_images) h->max_images = (int)(imgs > 1000000 ? 1000000 : imgs);
    }
    if (h->d_pos.ensure(sizeof(double) * 3 * N) || h->d_Z.ensure(sizeof(int) * N) ||
        h->d_atom_cfg.ensure(sizeof(int) * N) || h->d_cfg_start.ensure(sizeof(int) * (n_cfg + 1)) ||
        h->d_cell.ensure(sizeof(double) * 9 * n_cfg) || h->d_invcell.ensure(sizeof(double) * 9 * n_cfg) ||
        h->d_nimg.ensure(sizeof(int) * 3 * n_cfg) || h->d_pbc.ensure((size_t)3 * n_cfg))
        return set_err(h, VSSR_E_NOMEM, "batch buffers: out of device memory");
    VSSR_HIP(h, hipMemcpy(h->d_pos.p, pos, sizeof(double) * 3 * N, hipMemcpyHostToDevice));
    VSSR_HIP(h, hipMemcpy(h->d_Z.p, Z, sizeof(int) * N, hipMemcpyHostToDevice));
    VSSR_HIP(h, hipMemcpy(h->d_atom_cfg.p, atom_cfg.data(), sizeof(int) * N, hipMemcpyHostToDevice));
    VSSR_HIP(h, hipMemcpy(h->d_cfg_start.p, start.data(), sizeof(int) * (n_cfg + 1), hipMemcpyHostToDevice));
    VSSR_HIP(h, hipMemcpy(h->d_cell.p, cell, sizeof(double) * 9 * n_cfg, hipMemcpyHostToDevice));
    VSSR_HIP(h, hipMemcpy(h->d_invcell.p, inv.data(), sizeof(double) * 9 * n_cfg, hipMemcpyHostToDevice));
    VSSR_HIP(h, hipMemcpy(h->d_nimg.p, nimg.data(), sizeof(int) * 3 * n_cfg, hipMemcpyHostToDevice));
    VSSR_HIP(h, hipMemcpy(h->d_pbc.p, pbc, (size_t)3 * n_cfg, hipMemcpyHostToDevice));
    h->n_cfg = n_cfg;
    h->n_atoms = N;
    if (h->kind == 1) {   // species present in this batch (layer-0 factorisation works per neighbor species)
        std::vector<int> zmap(h->n_embed, -1), zlist;
        for (int i = 0; i < N; ++i)
            if (zmap[Z[i]] < 0) { zmap[Z[i]] = 1; }
        for (int z = 0; z < h->n_embed; ++z)
            if (zmap[z] > 0) { zmap[z] = (int)zlist.size(); zlist.push_back(z); }
        h->l0_nz = (int)zlist.size() <= L0_MAX_SPECIES ? (int)zlist.size() : 0;
        if (h->d_zmap.ensure(sizeof(int) * h->n_embed) || h->d_zlist.ensure(sizeof(int) * (zlist.size() + 1)))
            return set_err(h, VSSR_E_NOMEM, "species tables");
        VSSR_HIP(h, hipMemcpy(h->d_zmap.p, zmap.data(), sizeof(int) * h->n_embed, hipMemcpyHostToDevice));
        VSSR_HIP(h, hipMemcpy(h->d_zlist.p, zlist.data(), sizeof(int) * zlist.size(), hipMemcpyHostToDevice));
    }
    h->max_cfg_atoms = 0;
    for (int b = 0; b < n_cfg; ++b) h->max_cfg_atoms = n_atoms[b] > h->max_cfg_atoms ? n_atoms[b] : h->max_cfg_atoms;
    if (h->kind == 1) {
        // Neighbor-sum path of every chain, from its OWN atom count (so a chain's results do not depend on its batch):
        // 16-, 8- or 4-feature slices, or the gather kernels.  (A batch whose largest chain exceeds what the bundle
        // sort stages in LDS has no bundle tables at all: every chain gathers.)
        const bool bundles = (size_t)h->max_cfg_atoms * sizeof(int) <= 48 * 1024;
        std::vector<unsigned char> bcls(n_cfg);
        std::vector<int> lists[EDGE_MFMA_CLASSES], blists[EDGE_MFMA_BCLASSES];
        for (int c = 0; c < EDGE_CLASSES; ++c) { h->n_class[c] = 0; h->max_class_atoms[c] = 0; }
        for (int c = 0; c < EDGE_BCLASSES; ++c) { h->n_bclass[c] = 0; h->max_bclass_atoms[c] = 0; }
        for (int b = 0; b < n_cfg; ++b) {
            int c = (h->edge_impl && bundles) ? edge_class_of(n_atoms[b]) : EDGE_CLASS_GATHER;
            int bc = (h->edge_impl && bundles) ? edge_bclass_of(n_atoms[b]) : EDGE_BCLASS_GATHER;
            if (c <= EDGE_CLASS_FS16M && h->fs16_max_atoms >= 0 && n_atoms[b] > h->fs16_max_atoms) c = EDGE_CLASS_FS8;
            if (bc == EDGE_BCLASS_FS16 && h->fs16_max_atoms >= 0 && n_atoms[b] > h->fs16_max_atoms) bc = EDGE_BCLASS_FS8;
            if (c == EDGE_CLASS_FS8 && h->fs8_max_atoms >= 0 && n_atoms[b] > h->fs8_max_atoms) { c = EDGE_CLASS_FS4; bc = EDGE_BCLASS_FS4; }
            if (c == EDGE_CLASS_FS4 && h->fs4_max_atoms >= 0 && n_atoms[b] > h->fs4_max_atoms) { c = EDGE_CLASS_GATHER; bc = EDGE_BCLASS_GATHER; }
            // the 8-feature forward class (406 .. 787 atoms by the chain's own size) also takes the 16-feature multi-pass form
            if (h->fwd_mpass_fs8 && h->fwd_two_pass == 16 && c == EDGE_CLASS_FS8 && edge_class_of(n_atoms[b]) == EDGE_CLASS_FS8) c = EDGE_CLASS_FS4;
            // reverse pass of chains beyond the single-pass 16-feature form (by the chain's OWN size, not by a test knob that moved it):
            // the same kernel in several passes over neighbor sub-ranges
            if (bc != EDGE_BCLASS_GATHER && ((h->bwd_multi_pass == 1 && edge_bclass_of(n_atoms[b]) != EDGE_BCLASS_FS16) || h->bwd_multi_pass == 2))
                bc = EDGE_BCLASS_FS16P;
            bcls[b] = (unsigned char)bc;
            h->n_class[c] += 1;
            h->n_bclass[bc] += 1;
            if (n_atoms[b] > h->max_class_atoms[c]) h->max_class_atoms[c] = n_atoms[b];
            if (n_atoms[b] > h->max_bclass_atoms[bc]) h->max_bclass_atoms[bc] = n_atoms[b];
            if (c != EDGE_CLASS_GATHER) lists[c].push_back(b);
            if (bc != EDGE_BCLASS_GATHER) blists[bc].push_back(b);
        }
        std::vector<int> cat;
        for (int c = 0; c < EDGE_MFMA_CLASSES; ++c) cat.insert(cat.end(), lists[c].begin(), lists[c].end());
        for (int c = 0; c < EDGE_MFMA_BCLASSES; ++c) cat.insert(cat.end(), blists[c].begin(), blists[c].end());
        cat.push_back(0);
        const std::vector<unsigned char> &cls = bcls;
        if (h->d_chain_class.ensure((size_t)n_cfg) || h->d_class_list.ensure(sizeof(int) * cat.size()))
            return set_err(h, VSSR_E_NOMEM, "chain class tables");
        VSSR_HIP(h, hipMemcpy(h->d_chain_class.p, cls.data(), (size_t)n_cfg, hipMemcpyHostToDevice));
        VSSR_HIP(h, hipMemcpy(h->d_class_list.p, cat.data(), sizeof(int) * cat.size(), hipMemcpyHostToDevice));
    }
    h->h_n_atoms.assign(n_atoms, n_atoms + n_cfg);
    h->h_cfg_start = start;
    h->batch_valid = true;
    return VSSR_OK;
}

int vssr_batch_set_positions(vssr_handle *h, const double *pos) {
    if (!h) return VSSR_E_BADARG;
    if (!h->batch_valid) return set_err(h, VSSR_E_STATE, "no resident batch");
    if (!pos) return set_err(h, VSSR_E_BADARG, "null positions");
    VSSR_HIP(h, hipSetDevice(h->device));
    VSSR_HIP(h, hipStreamSynchronize(h->stream));
    VSSR_HIP(h, hipMemcpy(h->d_pos.p, pos, sizeof(double) * 3 * h->n_atoms, hipMemcpyHostToDevice));
    h->ran = false;   // results on the device belong to the old positions
    return VSSR_OK;
}

int vssr_batch_run(vssr_handle *h, uint32_t want) {
    if (!h) return VSSR_E_BADARG;
    if (!h->batch_valid) return set_err(h, VSSR_E_STATE, "vssr_batch_run before vssr_batch_upload");
    VSSR_HIP(h, hipSetDevice(h->device));
    int rc = run_any(h, want);
    if (rc) return rc;
    h->ran = true;
    h->graph_partial = false;
    return VSSR_OK;
}

int vssr_synchronize(vssr_handle *h) {
    if (!h) return VSSR_E_BADARG;
    VSSR_HIP(h, hipSetDevice(h->device));
    return sync_and_check(h);
}

int vssr_batch_download(vssr_handle *h, uint32_t want, vssr_out *out) {
    if (!h) return VSSR_E_BADARG;
    if (!h->ran) return set_err(h, VSSR_E_STATE, "vssr_batch_download before vssr_batch_run");
    if (!out) return set_err(h, VSSR_E_BADARG, "null output");
    VSSR_HIP(h, hipSetDevice(h->device));
    if ((want & VSSR_WANT_FORCES) && !(h->last_want & VSSR_WANT_FORCES))
        return set_err(h, VSSR_E_STATE, "forces requested, but the last run was asked for energies only");
    int rc = sync_and_check(h);
    if (rc) return rc;
    const size_t B = h->n_cfg, N = h->n_atoms, M = h->n_models;
    if (h->kind == 2 || h->kind == 3) {
        std::vector<double> e(B), ea(N), f(3 * N);
        VSSR_HIP(h, hipMemcpy(e.data(), h->d_ters_e.p, sizeof(double) * B, hipMemcpyDeviceToHost));
        if (out->energy) for (size_t b = 0; b < B; ++b) out->energy[b] = (float)e[b];
        if (out->energy_atoms && (want & VSSR_WANT_PER_ATOM)) {
            VSSR_HIP(h, hipMemcpy(ea.data(), h->d_ters_ea.p, sizeof(double) * N, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < N; ++i) out->energy_atoms[i] = (float)ea[i];
        }
        if (out->forces && (want & VSSR_WANT_FORCES)) {
            VSSR_HIP(h, hipMemcpy(f.data(), h->d_ters_f.p, sizeof(double) * 3 * N, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < 3 * N; ++i) out->forces[i] = (float)f[i];
        }
        return VSSR_OK;
    }
    if (out->energy) VSSR_HIP(h, hipMemcpy(out->energy, h->d_energy.p, sizeof(float) * B, hipMemcpyDeviceToHost));
    if (out->energy_std && (want & VSSR_WANT_STD))
        VSSR_HIP(h, hipMemcpy(out->energy_std, h->d_energy_std.p, sizeof(float) * B, hipMemcpyDeviceToHost));
    if (out->energy_models && (want & VSSR_WANT_PER_MODEL))
        VSSR_HIP(h, hipMemcpy(out->energy_models, h->d_energy_models.p, sizeof(float) * B * M, hipMemcpyDeviceToHost));
    if (out->energy_atoms && (want & VSSR_WANT_PER_ATOM))
        VSSR_HIP(h, hipMemcpy(out->energy_atoms, h->d_e_atoms.p, sizeof(float) * N, hipMemcpyDeviceToHost));
    if (want & VSSR_WANT_FORCES) {
        if (out->forces) VSSR_HIP(h, hipMemcpy(out->forces, h->d_forces.p, sizeof(float) * 3 * N, hipMemcpyDeviceToHost));
        if (out->forces_std && (want & VSSR_WANT_STD))
            VSSR_HIP(h, hipMemcpy(out->forces_std, h->d_forces_std.p, sizeof(float) * 3 * N, hipMemcpyDeviceToHost));
    }
    // the saturation report travels with the results (vssr_batch_saturated then needs no synchronisation / copy of its own)
    h->h_sat.resize(B);
    VSSR_HIP(h, hipMemcpy(h->h_sat.data(), h->d_sat_out.p, sizeof(unsigned) * B, hipMemcpyDeviceToHost));
    h->h_sat_valid = true;
    return VSSR_OK;
}

int vssr_eval_batch(vssr_handle *h, int32_t n_cfg, const int32_t *n_atoms, const int32_t *Z, const double *pos,
                    const double *cell, const uint8_t *pbc, uint32_t want, vssr_out *out) {
    int rc = vssr_batch_upload(h, n_cfg, n_atoms, Z, pos, cell, pbc);
    if (rc) return rc;
    rc = vssr_batch_run(h, want);
    if (rc) return rc;
    return vssr_batch_download(h, want, out);
}

int vssr_eval(vssr_handle *h, int32_t n_atoms, const int32_t *Z, const double *pos, const double cell[9],
              const uint8_t pbc[3], uint32_t want, vssr_out *out) {
    return vssr_eval_batch(h, 1, &n_atoms, Z, pos, cell, pbc, want, out);
}

int vssr_tersoff_eval_batch(vssr_handle *h, int32_t n_cfg, const int32_t *n_atoms, const int32_t *type,
                            const double *pos, const double *cell, const uint8_t *pbc, uint32_t want, vssr_out *out,
                            double *energy_f64, double *energy_atoms_f64, double *forces_f64) {
    if (!h) return VSSR_E_BADARG;
    if (h->kind != 2 && h->kind != 3) return set_err(h, VSSR_E_STATE, "not a Tersoff / EAM handle");
    vssr_out dummy;
    memset(&dummy, 0, sizeof dummy);
    int rc = vssr_eval_batch(h, n_cfg, n_atoms, type, pos, cell, pbc, want, out ? out : &dummy);
    if (rc) return rc;
    if (energy_f64) VSSR_HIP(h, hipMemcpy(energy_f64, h->d_ters_e.p, sizeof(double) * h->n_cfg, hipMemcpyDeviceToHost));
    if (energy_atoms_f64)
        VSSR_HIP(h, hipMemcpy(energy_atoms_f64, h->d_ters_ea.p, sizeof(double) * h->n_atoms, hipMemcpyDeviceToHost));
    if (forces_f64)
        VSSR_HIP(h, hipMemcpy(forces_f64, h->d_ters_f.p, sizeof(double) * 3 * h->n_atoms, hipMemcpyDeviceToHost));
    return VSSR_OK;
}

static int relax_finish(vssr_handle *h, double *pos_out, int32_t *n_steps, uint8_t *converged);

int vssr_batch_relax_cg(vssr_handle *h, const vssr_cg_params *params, const uint8_t *fixed, uint32_t want, double *pos_out,
                        int32_t *n_iter, int32_t *n_eval, int32_t *stop_reason) {
    if (!h) return VSSR_E_BADARG;
    if (!h->batch_valid) return set_err(h, VSSR_E_STATE, "vssr_batch_relax_cg before vssr_batch_upload");
    if (!params || params->max_iter < 0 || params->max_eval < 1 || !(params->etol >= 0) || !(params->ftol >= 0) || !(params->dmax > 0))
        return set_err(h, VSSR_E_BADARG, "bad CG parameters");
    VSSR_HIP(h, hipSetDevice(h->device));
    h->relax_regrows = 0;
    h->last_want = want | VSSR_WANT_FORCES;
    // chains of <= 256 atoms on the Tersoff potential: one workgroup minimises one chain from start to stop (chain_min.hip); else
    // the lock-step driver (relax.hip).  Same results bit for bit.
    const bool resident = chain_min_supported(h);
    int rc = resident ? chain_min_cg(h, params, fixed, want) : relax_cg(h, params, fixed, want);
    if (rc) return rc;
    h->graph_partial = resident;     // (the lock-step driver ends with a full batch-wide evaluation of the final positions; the
                                     //  chain-resident one numbers its rows per chain: introspection wants one plain run first)
    rc = sync_and_check(h);          // ... which may itself have overflowed the neighbor capacity: grow and repeat it
    if (rc) return rc;
    if (pos_out) VSSR_HIP(h, hipMemcpy(pos_out, h->d_pos.p, sizeof(double) * 3 * h->n_atoms, hipMemcpyDeviceToHost));
    std::vector<int> rep((size_t)3 * h->n_cfg);
    VSSR_HIP(h, hipMemcpy(rep.data(), h->d_relax_steps.p, sizeof(int) * rep.size(), hipMemcpyDeviceToHost));
    for (int b = 0; b < h->n_cfg; ++b) {
        if (n_iter) n_iter[b] = rep[3 * b];
        if (n_eval) n_eval[b] = rep[3 * b + 1];
        if (stop_reason) stop_reason[b] = rep[3 * b + 2];
    }
    return VSSR_OK;
}

int vssr_eam_eval_batch(vssr_handle *h, int32_t n_cfg, const int32_t *n_atoms, const int32_t *type, const double *pos,
                        const double *cell, const uint8_t *pbc, uint32_t want, vssr_out *out, double *energy_f64,
                        double *energy_atoms_f64, double *forces_f64) {
    return vssr_tersoff_eval_batch(h, n_cfg, n_atoms, type, pos, cell, pbc, want, out, energy_f64, energy_atoms_f64,
                                   forces_f64);
}

int vssr_batch_relax_bfgs(vssr_handle *h, const vssr_bfgs_params *params, const uint8_t *fixed, uint32_t want,
                          double *pos_out, int32_t *n_steps, uint8_t *converged) {
    if (!h) return VSSR_E_BADARG;
    if (!h->batch_valid) return set_err(h, VSSR_E_STATE, "vssr_batch_relax_bfgs before vssr_batch_upload");
    if (!params || params->max_steps < 0 || !(params->fmax > 0) || !(params->alpha > 0) || !(params->maxstep > 0))
        return set_err(h, VSSR_E_BADARG, "bad BFGS parameters");
    VSSR_HIP(h, hipSetDevice(h->device));
    h->relax_regrows = 0;
    h->last_want = want | VSSR_WANT_FORCES;
    int rc = relax_run(h, 1, nullptr, params, fixed, want);
    if (rc) return rc;
    return relax_finish(h, pos_out, n_steps, converged);
}

int vssr_batch_relax_fire(vssr_handle *h, const vssr_fire_params *params, const uint8_t *fixed, uint32_t want,
                          double *pos_out, int32_t *n_steps, uint8_t *converged) {
    if (!h) return VSSR_E_BADARG;
    if (!h->batch_valid) return set_err(h, VSSR_E_STATE, "vssr_batch_relax_fire before vssr_batch_upload");
    if (!params || params->max_steps < 0 || !(params->fmax > 0) || !(params->dt > 0) || !(params->maxstep > 0))
        return set_err(h, VSSR_E_BADARG, "bad FIRE parameters");
    VSSR_HIP(h, hipSetDevice(h->device));
    h->relax_regrows = 0;
    h->last_want = want | VSSR_WANT_FORCES;
    int rc = relax_run(h, 0, params, nullptr, fixed, want);
    if (rc) return rc;
    return relax_finish(h, pos_out, n_steps, converged);
}

static int relax_finish(vssr_handle *h, double *pos_out, int32_t *n_steps, uint8_t *converged) {
    VSSR_HIP(h, hipStreamSynchronize(h->stream));
    h->prof.collect();
    if (pos_out) VSSR_HIP(h, hipMemcpy(pos_out, h->d_pos.p, sizeof(double) * 3 * h->n_atoms, hipMemcpyDeviceToHost));
    if (n_steps) VSSR_HIP(h, hipMemcpy(n_steps, h->d_relax_steps.p, sizeof(int) * h->n_cfg, hipMemcpyDeviceToHost));
    if (converged) VSSR_HIP(h, hipMemcpy(converged, h->d_relax_conv.p, (size_t)h->n_cfg, hipMemcpyDeviceToHost));
    return VSSR_OK;
}

// ---- introspection ---------------------------------------------------------------------------------------
int vssr_profile_enable(vssr_handle *h, int enable) {
    if (!h) return VSSR_E_BADARG;
    VSSR_HIP(h, hipStreamSynchronize(h->stream));
    h->prof.collect();
    h->prof.enabled = enable != 0;
    return VSSR_OK;
}
int vssr_profile_reset(vssr_handle *h) {
    if (!h) return VSSR_E_BADARG;
    VSSR_HIP(h, hipStreamSynchronize(h->stream));
    h->prof.reset();
    return VSSR_OK;
}
int vssr_profile_read(vssr_handle *h, int32_t cap, const char **names, int64_t *launches, double *total_ms,
                      int32_t *n_out) {
    if (!h || !n_out) return VSSR_E_BADARG;
    VSSR_HIP(h, hipStreamSynchronize(h->stream));
    h->prof.collect();
    int n = 0;
    for (int k = 0; k < KC_COUNT && n < cap; ++k) {
        if (names) names[n] = kKernelClassNames[k];
        if (launches) launches[n] = h->prof.launches[k];
        if (total_ms) total_ms[n] = h->prof.total_ms[k];
        ++n;
    }
    *n_out = n;
    return VSSR_OK;
}

int vssr_batch_stats(vssr_handle *h, int64_t *n_atoms, int64_t *n_edges, int64_t *n_slots) {
    if (!h) return VSSR_E_BADARG;
    if (!h->ran) return set_err(h, VSSR_E_STATE, "no completed run");
    if (h->graph_partial)
        return set_err(h, VSSR_E_STATE, "the resident graph covers only the chains of the last relaxation iteration: run the batch once (vssr_batch_run) first");
    int rc = sync_and_check(h);
    if (rc) return rc;
    if (n_atoms) *n_atoms = h->n_atoms;
    if (n_edges) *n_edges = h->h_counters[1];
    if (n_slots) *n_slots = h->h_counters[0];
    return VSSR_OK;
}

int vssr_batch_neighbors(vssr_handle *h, int64_t cap, int32_t *ei, int32_t *ej, int32_t *eS, float *er,
                         int64_t *n_edges) {
    if (!h || !n_edges) return VSSR_E_BADARG;
    if (!h->ran) return set_err(h, VSSR_E_STATE, "no completed run");
    if (h->graph_partial)
        return set_err(h, VSSR_E_STATE, "the resident graph covers only the chains of the last relaxation iteration: run the batch once (vssr_batch_run) first");
    int rc = sync_and_check(h);
    if (rc) return rc;
    const int N = h->n_atoms;
    const int64_t slots = h->h_counters[0];
    *n_edges = h->h_counters[1];
    if (!ei && !ej && !eS && !er) return VSSR_OK;
    std::vector<int> row(N + 1), S(slots), wrap((size_t)3 * N);
    std::vector<float4> edge(slots);
    VSSR_HIP(h, hipMemcpy(row.data(), h->d_row_start.p, sizeof(int) * (N + 1), hipMemcpyDeviceToHost));
    VSSR_HIP(h, hipMemcpy(S.data(), h->d_edge_S.p, sizeof(int) * slots, hipMemcpyDeviceToHost));
    VSSR_HIP(h, hipMemcpy(edge.data(), h->d_edge.p, sizeof(float4) * slots, hipMemcpyDeviceToHost));
    VSSR_HIP(h, hipMemcpy(wrap.data(), h->d_wrap.p, sizeof(int) * 3 * N, hipMemcpyDeviceToHost));
    int64_t n = 0;
    for (int i = 0; i < N; ++i)
        for (int e = row[i]; e < row[i + 1]; ++e) {
            int j;
            memcpy(&j, &edge[e].w, sizeof(int));
            if (j < 0) continue;
            if (n < cap) {
                if (ei) ei[n] = i;
                if (ej) ej[n] = j;
                if (eS)  // true image shift: S = S' + wrap_i - wrap_j
                    for (int k = 0; k < 3; ++k)
                        eS[3 * n + k] = (((S[e] >> (8 * k)) & 255) - 128) + wrap[3 * i + k] - wrap[3 * j + k];
                if (er) { er[3 * n] = edge[e].x; er[3 * n + 1] = edge[e].y; er[3 * n + 2] = edge[e].z; }
            }
            ++n;
        }
    return VSSR_OK;
}

int vssr_batch_device_results(vssr_handle *h, const float **energy, const float **energy_std) {
    if (!h) return VSSR_E_BADARG;
    if (!h->ran || h->kind != 1) return set_err(h, VSSR_E_STATE, "no completed PaiNN run");
    if (energy) *energy = h->d_energy.as<float>();
    if (energy_std) *energy_std = h->d_energy_std.as<float>();
    return VSSR_OK;
}

int vssr_batch_device_results_f64(vssr_handle *h, const double **energy, const double **energy_std) {
    if (!h) return VSSR_E_BADARG;
    if (!h->ran || h->kind != 1) return set_err(h, VSSR_E_STATE, "no completed PaiNN run");
    if (energy) *energy = h->d_energy64.as<double>();
    if (energy_std) *energy_std = h->d_energy64.as<double>() + h->n_cfg;
    return VSSR_OK;
}

int vssr_batch_energy_f64(vssr_handle *h, double *energy, double *energy_std, double *energy_models) {
    if (!h) return VSSR_E_BADARG;
    if (!h->ran) return set_err(h, VSSR_E_STATE, "vssr_batch_energy_f64 before a run");
    VSSR_HIP(h, hipSetDevice(h->device));
    int rc = sync_and_check(h);
    if (rc) return rc;
    const size_t B = h->n_cfg, M = h->n_models;
    if (h->kind == 2 || h->kind == 3) {   // one analytic potential: no spread, the "model" is the potential
        if (energy) VSSR_HIP(h, hipMemcpy(energy, h->d_ters_e.p, sizeof(double) * B, hipMemcpyDeviceToHost));
        if (energy_models) VSSR_HIP(h, hipMemcpy(energy_models, h->d_ters_e.p, sizeof(double) * B, hipMemcpyDeviceToHost));
        if (energy_std) for (size_t b = 0; b < B; ++b) energy_std[b] = 0.0;
        return VSSR_OK;
    }
    const double *src = h->d_energy64.as<double>();
    if (energy) VSSR_HIP(h, hipMemcpy(energy, src, sizeof(double) * B, hipMemcpyDeviceToHost));
    if (energy_std) VSSR_HIP(h, hipMemcpy(energy_std, src + B, sizeof(double) * B, hipMemcpyDeviceToHost));
    if (energy_models) VSSR_HIP(h, hipMemcpy(energy_models, src + 2 * B, sizeof(double) * B * M, hipMemcpyDeviceToHost));
    return VSSR_OK;
}

int vssr_device_context(vssr_handle *h, int32_t *device, void **stream, const int32_t **overflow_flag) {
    if (!h) return VSSR_E_BADARG;
    if (device) *device = h->device;
    if (stream) *stream = (void *)h->stream;
    if (overflow_flag) *overflow_flag = h->d_counters.p ? h->d_counters.as<int>() + 2 : nullptr;
    return VSSR_OK;
}

int vssr_batch_traj_configure(vssr_handle *h, int32_t record_interval) {
    if (!h) return VSSR_E_BADARG;
    if (record_interval < 0) return set_err(h, VSSR_E_BADARG, "record_interval must be >= 0");
    h->traj_interval = record_interval;
    return VSSR_OK;
}

int vssr_batch_traj_read(vssr_handle *h, int32_t cap_records, int32_t *n_records, double *pos, float *forces, double *energy,
                         int32_t *max_records) {
    if (!h) return VSSR_E_BADARG;
    if (max_records) *max_records = h->traj_records;
    if (!h->traj_records) {
        if (n_records || pos || forces || energy) return set_err(h, VSSR_E_STATE, "the last relaxation recorded no trajectory");
        return VSSR_OK;
    }
    if (!n_records && !pos && !forces && !energy) return VSSR_OK;
    if (h->traj_B != h->n_cfg || h->traj_N != h->n_atoms) return set_err(h, VSSR_E_STATE, "the recorded trajectory belongs to another batch");
    if (cap_records < h->traj_records) return set_err(h, VSSR_E_BADARG, "trajectory buffers hold %d records, %d are needed", cap_records, h->traj_records);
    VSSR_HIP(h, hipSetDevice(h->device));
    VSSR_HIP(h, hipStreamSynchronize(h->stream));
    const size_t R = h->traj_records, N3 = (size_t)3 * h->traj_N, B = h->traj_B;
    if (n_records) VSSR_HIP(h, hipMemcpy(n_records, h->d_traj_n.p, sizeof(int) * B, hipMemcpyDeviceToHost));
    if (pos) VSSR_HIP(h, hipMemcpy(pos, h->d_traj_pos.p, sizeof(double) * N3 * R, hipMemcpyDeviceToHost));
    if (forces) VSSR_HIP(h, hipMemcpy(forces, h->d_traj_f.p, sizeof(float) * N3 * R, hipMemcpyDeviceToHost));
    if (energy) VSSR_HIP(h, hipMemcpy(energy, h->d_traj_e.p, sizeof(double) * B * R, hipMemcpyDeviceToHost));
    return VSSR_OK;
}

int vssr_batch_embedding(vssr_handle *h, int32_t model, float *dst, int64_t cap, int64_t *n_out) {
    if (!h) return VSSR_E_BADARG;
    if (!h->ran || h->kind != 1) return set_err(h, VSSR_E_STATE, "no completed PaiNN run");
    if (model < -1 || model >= h->n_models) return set_err(h, VSSR_E_BADARG, "model index out of range");
    if (h->graph_partial)   // (chains that converged early keep the features of THEIR last iteration, or of a buffer a regrow replaced)
        return set_err(h, VSSR_E_STATE, "the resident activations cover only the chains of the last relaxation iteration: run the batch once (vssr_batch_run) first");
    VSSR_HIP(h, hipSetDevice(h->device));
    int rc = sync_and_check(h);
    if (rc) return rc;
    const size_t per_model = (size_t)h->n_atoms * F, n = model < 0 ? per_model * h->n_models : per_model;
    if (n_out) *n_out = (int64_t)n;
    if (!dst) return VSSR_OK;
    if ((int64_t)n > cap) return set_err(h, VSSR_E_BADARG, "embedding buffer too small (%lld < %zu floats)", (long long)cap, n);
    // the scalar features leaving the last update block: the state the readout consumes ([M][N][F], model-major)
    const float *src = h->sv.s_in[h->num_conv] + (model < 0 ? 0 : (size_t)model * per_model);
    VSSR_HIP(h, hipMemcpy(dst, src, n * sizeof(float), hipMemcpyDeviceToHost));
    return VSSR_OK;
}

int vssr_batch_stress(vssr_handle *h, double *stress, double *stress_std) {
    if (!h) return VSSR_E_BADARG;
    if (!h->ran || h->kind != 1) return set_err(h, VSSR_E_STATE, "no completed PaiNN run");
    if (!(h->last_want & VSSR_WANT_FORCES))
        return set_err(h, VSSR_E_STATE, "stress needs the edge gradients of a run that produced forces; the last run was asked for energies only");
    if (h->graph_partial)
        return set_err(h, VSSR_E_STATE, "the resident graph covers only the chains of the last relaxation iteration: run the batch once (vssr_batch_run) first");
    VSSR_HIP(h, hipSetDevice(h->device));
    int rc = sync_and_check(h);   // (a capacity overflow is repaired here: the gradients below are those of the repeated run)
    if (rc) return rc;
    rc = painn_stress(h);
    if (rc) return rc;
    VSSR_HIP(h, hipStreamSynchronize(h->stream));
    const size_t n = 6 * (size_t)h->n_cfg;
    if (stress) VSSR_HIP(h, hipMemcpy(stress, h->d_stress.p, sizeof(double) * n, hipMemcpyDeviceToHost));
    if (stress_std) VSSR_HIP(h, hipMemcpy(stress_std, h->d_stress.as<double>() + n, sizeof(double) * n, hipMemcpyDeviceToHost));
    return VSSR_OK;
}

int vssr_batch_saturated(vssr_handle *h, uint8_t *flags, int32_t *n_flagged) {
    if (!h) return VSSR_E_BADARG;
    if (!h->ran) return set_err(h, VSSR_E_STATE, "no completed run");
    VSSR_HIP(h, hipSetDevice(h->device));
    int count = 0;
    if (h->kind == 1) {   // the fp64 potentials have no reduced-precision stage
        if (!h->h_sat_valid || (int)h->h_sat.size() != h->n_cfg) {
            int rc = sync_and_check(h);
            if (rc) return rc;
            h->h_sat.resize(h->n_cfg);
            VSSR_HIP(h, hipMemcpy(h->h_sat.data(), h->d_sat_out.p, sizeof(unsigned) * h->n_cfg, hipMemcpyDeviceToHost));
            h->h_sat_valid = true;
        }
        const std::vector<unsigned> &f = h->h_sat;
        for (int b = 0; b < h->n_cfg; ++b) {
            if (flags) flags[b] = f[b] ? 1 : 0;
            count += f[b] ? 1 : 0;
        }
    } else if (flags) {
        memset(flags, 0, (size_t)h->n_cfg);
    }
    if (n_flagged) *n_flagged = count;
    return VSSR_OK;
}

int vssr_debug_capacity(vssr_handle *h, int32_t slots_per_atom, int32_t tight, int32_t *n_regrows) {
    if (!h) return VSSR_E_BADARG;
    if (slots_per_atom > 0) {
        h->cap_per_atom = slots_per_atom;
        h->cm_cap_per_atom = 0;   // (the chain-resident minimiser's pools start from the new value as well)
        h->slot_cap = 0;   // re-derived at the next neighbor build
    }
    if (tight >= 0) h->cap_tight = tight != 0;
    if (n_regrows) *n_regrows = h->relax_regrows;
    return VSSR_OK;
}

int vssr_batch_relax_counts(vssr_handle *h, int64_t *lockstep_evaluations, int64_t *chain_evaluations) {
    if (!h) return VSSR_E_BADARG;
    if (lockstep_evaluations) *lockstep_evaluations = h->relax_lockstep;
    if (chain_evaluations) *chain_evaluations = h->relax_chain_evals;
    return VSSR_OK;
}

int vssr_debug_read(vssr_handle *h, const char *name, int32_t model, float *dst, int64_t cap, int64_t *n_out) {
    if (!h || !name || !n_out) return VSSR_E_BADARG;
    if (!h->ran || h->kind != 1) return set_err(h, VSSR_E_STATE, "no completed PaiNN run");
    if (model < 0 || model >= h->n_models) return set_err(h, VSSR_E_BADARG, "model index out of range");
    if (h->graph_partial)
        return set_err(h, VSSR_E_STATE, "the resident graph covers only the chains of the last relaxation iteration: run the batch once (vssr_batch_run) first");
    int rc = sync_and_check(h);
    if (rc) return rc;
    const size_t N = h->n_atoms;
    const StateView &sv = h->sv;
    const float *src = nullptr;
    size_t per_atom = 0;
    std::string nm(name);
    auto layer_of = [&](const char *prefix) -> int {
        size_t pl = strlen(prefix);
        if (nm.compare(0, pl, prefix) != 0 || nm.size() != pl + 1) return -1;
        int l = nm[pl] - '0';
        return (l >= 0 && l < h->num_conv) ? l : -1;
    };
    int l;
    if ((l = layer_of("phi")) >= 0) {
        if (l == 0 && h->l0_used)
            return set_err(h, VSSR_E_STATE, "phi0 is not materialised (layer-0 species factorisation is active)");
        src = sv.phi[l]; per_atom = F3;
    }
    else if ((l = layer_of("s_msg")) >= 0) { src = sv.s_msg[l]; per_atom = F; }
    else if ((l = layer_of("v_msg")) >= 0) { src = sv.v_msg[l]; per_atom = F3; }
    else if ((l = layer_of("s_upd")) >= 0) { src = sv.s_in[l + 1]; per_atom = F; }
    else if ((l = layer_of("v_upd")) >= 0) {
        if (l == h->num_conv - 1 && !h->debug_keep)
            return set_err(h, VSSR_E_STATE, "v_upd of the last block is not materialised (nothing consumes it; create the handle with VSSR_DEBUG_KEEP=1)");
        src = sv.v_in[l + 1]; per_atom = F3;
    }
    else if (nm == "e_atom") { src = sv.e_atom; per_atom = 1; }
    else if (nm == "sbar_msg0") { src = sv.sbar_msg_l0; per_atom = F; }   // reverse buffers hold the LAST layer processed
    else if (nm == "vbar_msg0") { src = sv.vbar_msg; per_atom = F3; }
    else return set_err(h, VSSR_E_BADARG, "unknown intermediate '%s'", name);
    size_t n = N * per_atom;
    *n_out = (int64_t)n;
    if (!dst) return VSSR_OK;
    if ((int64_t)n > cap) return set_err(h, VSSR_E_BADARG, "buffer too small for '%s'", name);
    VSSR_HIP(h, hipMemcpy(dst, src + (size_t)model * n, n * sizeof(float), hipMemcpyDeviceToHost));
    return VSSR_OK;
}

}  // extern "C"

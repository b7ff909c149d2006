// libimx core: error plumbing, version, plan parsing/validation/upload.
//
// The plan is the compiled form of the reference's manager term lists
// (ManagerBase._prepare_terms, reference isaaclab/managers/manager_base.py:160 and the per-manager overrides):
// plan.py resolves names/regexes on the host and emits int32 words; here they are validated once so that the
// kernels can index without bounds checks.
#include <cstring>
#include <memory>

#include "imx_internal.h"

static thread_local std::string g_err;

void imx_set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
}

extern "C" const char* imx_version(void) { return "libimx 0.3 (gfx950; plan v3)"; }
extern "C" const char* imx_last_error(void) { return g_err.c_str(); }
extern "C" size_t imx_struct_size(int which) {
    switch (which) {
        case 0: return sizeof(imx_state_t);
        case 1: return sizeof(imx_buffers_t);
        case 2: return sizeof(imx_head_loss_t);
        case 3: return sizeof(imx_rollout_slot_t);
        case 4: return sizeof(imx_policy_act_t);
        case 5: return sizeof(imx_orch_t);
        case 6: return sizeof(imx_event_term_t);
        default: return 0;
    }
}

extern "C" int imx_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

static inline float wf(int32_t w) {
    float f;
    memcpy(&f, &w, 4);
    return f;
}

static int check_ids(const std::vector<int32_t>& w, int off, int n, int limit, const char* what, int rec) {
    if (n == 0) return 0;
    IMX_REQUIRE(off >= IMX_HEADER_WORDS && n > 0 && (size_t)off + (size_t)n <= w.size(),
                "plan: %s list of record %d out of range (off=%d n=%d)", what, rec, off, n);
    for (int i = 0; i < n; ++i)
        IMX_REQUIRE(w[off + i] >= 0 && w[off + i] < limit, "plan: %s index %d of record %d outside [0,%d)", what,
                    w[off + i], rec, limit);
    return 0;
}

// Validate a blob and expand it (column tables) into `p` (host side only).
static int parse_plan(const int32_t* blob, size_t nwords, imx_plan* p) {
    IMX_REQUIRE(nwords >= IMX_HEADER_WORDS, "plan: blob too small (%zu words)", nwords);
    IMX_REQUIRE(blob[IMX_H_MAGIC] == IMX_MAGIC, "plan: bad magic 0x%x", blob[IMX_H_MAGIC]);
    IMX_REQUIRE(blob[IMX_H_VERSION] == IMX_PLAN_VERSION, "plan: version %d, library expects %d", blob[IMX_H_VERSION],
                IMX_PLAN_VERSION);
    IMX_REQUIRE((size_t)blob[IMX_H_TOTAL_WORDS] == nwords, "plan: total_words %d != %zu", blob[IMX_H_TOTAL_WORDS],
                nwords);
    p->host.assign(blob, blob + nwords);
    auto& w = p->host;
    p->J = w[IMX_H_J]; p->B = w[IMX_H_B]; p->H = w[IMX_H_H]; p->A = w[IMX_H_A]; p->D = w[IMX_H_D];
    p->R = w[IMX_H_R]; p->NB = w[IMX_H_NB]; p->CMD = w[IMX_H_CMD_DIM];
    IMX_REQUIRE(w[IMX_H_MOD_STATE] >= 0 && w[IMX_H_MOD_STATE] < (1 << 20), "plan: modifier state width %d", w[IMX_H_MOD_STATE]);
    p->nterm = w[IMX_H_NTERM]; p->nrew = w[IMX_H_NREW]; p->nobs = w[IMX_H_NOBS]; p->nact = w[IMX_H_NACT];
    p->nrew_all = w[IMX_H_NREW_ALL];
    p->term_off = w[IMX_H_TERM_OFF]; p->rew_off = w[IMX_H_REW_OFF]; p->obs_off = w[IMX_H_OBS_OFF];
    p->act_off = w[IMX_H_ACT_OFF]; p->ray_off = w[IMX_H_RAY_OFF];
    IMX_REQUIRE(p->J >= 0 && p->J <= 4096 && p->B >= 0 && p->B <= 4096 && p->H >= 0 && p->H <= 64, "plan: bad J/B/H");
    IMX_REQUIRE(p->A >= 0 && p->A <= 4096 && p->D >= 0 && p->D <= 65535 && p->R >= 0 && p->R <= 65535, "plan: bad A/D/R");
    IMX_REQUIRE(p->nterm >= 0 && p->nterm <= 32, "plan: at most 32 termination terms are supported (got %d)", p->nterm);
    IMX_REQUIRE(p->nrew >= 0 && p->nrew == p->nrew_all && p->nrew_all <= 256, "plan: bad reward term counts (every term carries a record)");
    p->ngroups = w[IMX_H_NGROUPS];
    IMX_REQUIRE(p->ngroups >= 1 && p->ngroups <= IMX_MAX_OBS_GROUPS, "plan: %d observation groups (1..%d supported)", p->ngroups, IMX_MAX_OBS_GROUPS);
    IMX_REQUIRE(w[IMX_H_GROUP_OFF] >= IMX_HEADER_WORDS && (size_t)w[IMX_H_GROUP_OFF] + 4 * (size_t)p->ngroups <= nwords, "plan: group table out of range");
    IMX_REQUIRE(p->nobs >= 0 && p->nobs <= 4096 && p->nact >= 0 && p->nact <= 64, "plan: bad obs/action term counts");
    IMX_REQUIRE(p->CMD >= 0 && p->CMD <= 16, "plan: bad command dim");
    auto table_ok = [&](int off, int n) {
        return n == 0 || (off >= IMX_HEADER_WORDS && (size_t)off + (size_t)n * IMX_REC_WORDS <= nwords);
    };
    IMX_REQUIRE(table_ok(p->term_off, p->nterm) && table_ok(p->rew_off, p->nrew) && table_ok(p->obs_off, p->nobs) &&
                    table_ok(p->act_off, p->nact),
                "plan: record table out of range");
    if (p->R > 0)
        IMX_REQUIRE(p->ray_off >= IMX_HEADER_WORDS && (size_t)p->ray_off + (size_t)p->R * 3 <= nwords,
                    "plan: ray table out of range");

    // ---- terminations
    for (int k = 0; k < p->nterm; ++k) {
        const int32_t* r = &w[p->term_off + k * IMX_REC_WORDS];
        const int op = r[IMX_R_OP];
        IMX_REQUIRE(op >= IMX_T_TIME_OUT && op <= IMX_T_COMMAND_RESAMPLE, "plan: unknown termination op %d", op);
        IMX_REQUIRE(r[IMX_R_OUT] == k, "plan: termination record %d has index %d", k, r[IMX_R_OUT]);
        const bool body = op == IMX_T_ILLEGAL_CONTACT;
        if (op != IMX_T_COMMAND_RESAMPLE &&  // its NIDS word is num_resamples, not a list length
            check_ids(w, r[IMX_R_IDS_OFF], r[IMX_R_NIDS], body ? p->B : p->J, body ? "body" : "joint", k)) return 1;
        if (op == IMX_T_EXTERNAL)
            IMX_REQUIRE(r[IMX_R_AUX0] >= 0 && r[IMX_R_AUX0] < w[IMX_H_NEXT_TERM], "plan: ext_term column out of range");
    }
    // ---- rewards
    for (int k = 0; k < p->nrew; ++k) {
        const int32_t* r = &w[p->rew_off + k * IMX_REC_WORDS];
        const int op = r[IMX_R_OP];
        IMX_REQUIRE(op >= IMX_W_IS_ALIVE && op <= IMX_W_BODY_LIN_ACC_L2, "plan: unknown reward op %d", op);
        IMX_REQUIRE(r[IMX_R_OUT] == k, "plan: reward record %d has index %d", k, r[IMX_R_OUT]);
        int limit = p->J;
        const char* what = "joint";
        if (op == IMX_W_UNDESIRED_CONTACTS || op == IMX_W_CONTACT_FORCES || op == IMX_W_FEET_AIR_TIME ||
            op == IMX_W_FEET_AIR_TIME_POSITIVE_BIPED || op == IMX_W_FEET_SLIDE) {
            limit = p->B;
            what = "body";
        } else if (op == IMX_W_IS_TERMINATED_TERM) {
            limit = p->nterm;
            what = "termination-term";
        } else if (op == IMX_W_BODY_LIN_ACC_L2) {
            limit = p->NB;
            what = "asset body";
        }
        if (check_ids(w, r[IMX_R_IDS_OFF], r[IMX_R_NIDS], limit, what, k)) return 1;
        if (op == IMX_W_FEET_SLIDE) {
            IMX_REQUIRE(r[IMX_R_NIDS2] == r[IMX_R_NIDS], "plan: feet_slide needs equally many sensor and asset bodies");
            if (check_ids(w, r[IMX_R_IDS2_OFF], r[IMX_R_NIDS2], p->NB, "asset body", k)) return 1;
        }
        if (op == IMX_W_EXTERNAL)
            IMX_REQUIRE(r[IMX_R_AUX0] >= 0 && r[IMX_R_AUX0] < w[IMX_H_NEXT_REW], "plan: ext_reward column out of range");
    }
    // ---- observations: every column of every group covered exactly once.  Column space = the groups side by side
    //      (group g owns [gbase[g], gbase[g] + gD[g])); a record's OUT is relative to its group
    int tot = 0;
    for (int g = 0; g < p->ngroups; ++g) {
        const int32_t* gt = &w[w[IMX_H_GROUP_OFF] + 4 * g];
        IMX_REQUIRE(gt[0] > 0 && gt[0] <= 65535 && gt[2] >= 0 && gt[3] >= 0 && gt[2] + gt[3] <= p->nobs, "plan: bad observation group %d", g);
        p->gD[g] = gt[0];
        p->gbase[g] = tot;
        p->gcorrupt[g] = gt[1] != 0;
        tot += gt[0];
    }
    IMX_REQUIRE(tot == p->D, "plan: the observation groups cover %d columns, D=%d", tot, p->D);
    std::vector<int32_t> col(p->D, -1);
    std::vector<int> twin_of(p->nobs, -1);  // HEIGHT_SCAN record that shares the rays of an earlier one
    for (int k = 0; k < p->nobs; ++k) {
        const int32_t* r = &w[p->obs_off + k * IMX_REC_WORDS];
        const int op = r[IMX_R_OP];
        IMX_REQUIRE(op >= IMX_O_BASE_POS_Z && op <= IMX_O_EXTERNAL, "plan: unknown observation op %d", op);
        const int g = r[IMX_R_WEIGHT];
        IMX_REQUIRE(g >= 0 && g < p->ngroups, "plan: observation record %d names group %d", k, g);
        {
            const int32_t* gt = &w[w[IMX_H_GROUP_OFF] + 4 * g];
            IMX_REQUIRE(k >= gt[2] && k < gt[2] + gt[3], "plan: observation record %d outside the record range of its group %d", k, g);
        }
        const int o = r[IMX_R_OUT], d = r[IMX_R_DIM];
        // history (ObservationTermCfg.history_length, AUX1): the record owns hist*d columns, oldest slot first; only the newest
        // slot is computed, the older ones are shifted inside the obs buffer by the kernel
        const int hist = op == IMX_O_EXTERNAL ? 1 : (r[IMX_R_AUX1] > 1 ? r[IMX_R_AUX1] : 1);
        IMX_REQUIRE(op == IMX_O_EXTERNAL || (r[IMX_R_AUX1] >= 0 && r[IMX_R_AUX1] <= 64), "plan: observation record %d history length %d", k,
                    r[IMX_R_AUX1]);
        IMX_REQUIRE(o >= 0 && d > 0 && o + hist * d <= p->gD[g], "plan: observation record %d columns [%d,%d) outside D=%d of group %d", k, o,
                    o + hist * d, p->gD[g], g);
        const bool joint = op == IMX_O_JOINT_POS || op == IMX_O_JOINT_POS_REL || op == IMX_O_JOINT_VEL ||
                           op == IMX_O_JOINT_VEL_REL || op == IMX_O_JOINT_POS_LIMIT_NORMALIZED;
        if (joint) {
            IMX_REQUIRE(r[IMX_R_NIDS] == d, "plan: joint observation dim %d != number of joint ids %d", d, r[IMX_R_NIDS]);
            if (check_ids(w, r[IMX_R_IDS_OFF], r[IMX_R_NIDS], p->J, "joint", k)) return 1;
        }
        if (op == IMX_O_HEIGHT_SCAN) {
            IMX_REQUIRE(d == p->R && p->R > 0, "plan: height_scan dim %d != number of rays %d", d, p->R);
            p->needs_mesh = true;
            if (r[IMX_R_FLAGS] & IMX_F_SCAN_TWIN) {
                const int pr = r[IMX_R_AUX0];
                IMX_REQUIRE(pr >= 0 && pr < k && w[p->obs_off + pr * IMX_REC_WORDS + IMX_R_OP] == IMX_O_HEIGHT_SCAN &&
                                !(w[p->obs_off + pr * IMX_REC_WORDS + IMX_R_FLAGS] & IMX_F_SCAN_TWIN) && hist == 1,
                            "plan: observation record %d: bad height-scan twin reference %d", k, pr);
                twin_of[k] = pr;
            }
        } else {
            IMX_REQUIRE(!(r[IMX_R_FLAGS] & IMX_F_SCAN_TWIN), "plan: observation record %d: twin flag on a non-scan term", k);
        }
        if (op == IMX_O_LAST_ACTION) IMX_REQUIRE(d == p->A, "plan: last_action dim %d != A=%d", d, p->A);
        if (op == IMX_O_GENERATED_COMMANDS) IMX_REQUIRE(d == p->CMD, "plan: command dim mismatch");
        if (op == IMX_O_EXTERNAL)
            IMX_REQUIRE(r[IMX_R_AUX0] >= 0 && r[IMX_R_AUX0] + d <= w[IMX_H_NEXT_OBS], "plan: ext_obs columns out of range");
        if (r[IMX_R_FLAGS] & IMX_F_MODIFIERS) {  // modifier program: bounds, op codes, state slots inside the row
            const int po = r[IMX_R_IDS2_OFF], pn = r[IMX_R_NIDS2], so = r[IMX_R_P1];
            IMX_REQUIRE(po >= IMX_HEADER_WORDS && pn > 0 && (size_t)po + pn <= nwords, "plan: observation %d: modifier program out of range", k);
            int slots = 0;
            for (int q = 0; q < pn;) {
                IMX_REQUIRE(q + 4 <= pn, "plan: observation %d: truncated modifier program", k);
                const int mop = w[po + q];
                IMX_REQUIRE(mop >= IMX_M_SCALE && mop <= IMX_M_DIGITAL_FILTER, "plan: observation %d: unknown modifier op %d", k, mop);
                int need = 0, len = 4;
                if (mop == IMX_M_INTEGRATOR) need = 2;
                if (mop == IMX_M_DIGITAL_FILTER) {
                    const int na = w[po + q + 1], nb = w[po + q + 2];
                    IMX_REQUIRE(na >= 1 && nb >= 1 && na <= 64 && nb <= 64, "plan: observation %d: filter orders %d/%d", k, na, nb);
                    need = na + nb; len = 4 + na + nb;
                    IMX_REQUIRE(q + len <= pn, "plan: observation %d: truncated filter coefficients", k);
                }
                if (need) {
                    IMX_REQUIRE(w[po + q + 3] == slots, "plan: observation %d: modifier state slots must be consecutive", k);
                    slots += need;
                }
                q += len;
            }
            const int MSw = w[IMX_H_MOD_STATE];
            IMX_REQUIRE(so >= 0 && so + slots * d <= MSw, "plan: observation %d: modifier state [%d, %d) outside the %d-float row", k, so, so + slots * d, MSw);
        }
        static const int fixed_dim[] = {0, 1, 3, 3, 3, 3, 4, 3, 3};
        if (op <= IMX_O_ROOT_ANG_VEL_W) IMX_REQUIRE(d == fixed_dim[op], "plan: observation op %d must have dim %d", op, fixed_dim[op]);
        for (int h = 0; h < hist; ++h)
            for (int j = 0; j < d; ++j) {
                const int c = p->gbase[g] + o + h * d + j;
                IMX_REQUIRE(col[c] == -1, "plan: observation column %d written twice", c);
                col[c] = h == hist - 1 ? ((k << 16) | j) : -2;  // -2: an older history slot (not computed)
            }
    }
    for (int c = 0; c < p->D; ++c) IMX_REQUIRE(col[c] != -1, "plan: observation column %d not covered", c);
    // height scanner as a SensorBase
    p->scan_stateful = w[IMX_H_SCAN_STATEFUL] != 0;
    if (p->scan_stateful) {
        IMX_REQUIRE(p->R > 0 && w[IMX_H_SCAN_SUBSTEPS] >= 1 && w[IMX_H_SCAN_SUBSTEPS] <= 1024 && wf(w[IMX_H_SCAN_DT]) > 0.0f &&
                        wf(w[IMX_H_SCAN_PERIOD]) >= 0.0f && wf(w[IMX_H_SCAN_DRIFT_LO]) <= wf(w[IMX_H_SCAN_DRIFT_HI]),
                    "plan: bad height-scanner update period / drift words");
    }
    // ---- actions
    int acols = 0;
    for (int k = 0; k < p->nact; ++k) {
        const int32_t* r = &w[p->act_off + k * IMX_REC_WORDS];
        IMX_REQUIRE(r[IMX_R_OP] == IMX_A_JOINT_AFFINE, "plan: unknown action op %d", r[IMX_R_OP]);
        IMX_REQUIRE(r[IMX_R_OUT] == acols && r[IMX_R_DIM] == r[IMX_R_NIDS], "plan: action record %d layout", k);
        if (check_ids(w, r[IMX_R_IDS_OFF], r[IMX_R_NIDS], p->J, "joint", k)) return 1;
        // optional per-joint scale / offset / clip tables (dim floats each) referenced by aux words
        acols += r[IMX_R_DIM];
    }
    IMX_REQUIRE(acols == p->A || p->nact == 0, "plan: action terms cover %d columns, A=%d", acols, p->A);

    // ---- append column tables: [col (D)] [order (D)] with ray columns first; columns of twin height-scan records are not
    //      scheduled on their own (the lane of the primary ray finishes them)
    p->col_off = (int)w.size();
    w.insert(w.end(), col.begin(), col.end());
    p->order_off = (int)w.size();
    std::vector<int32_t> order, twins;
    order.reserve(p->D);
    p->n_ray_cols = 0;
    auto rec_of = [&](int c) { return col[c] >> 16; };
    for (int pass = 0; pass < 2; ++pass)
        for (int c = 0; c < p->D; ++c) {
            if (col[c] < 0) continue;  // history slot
            const int k = rec_of(c);
            if (twin_of[k] >= 0) { if (pass == 0) twins.push_back(c); continue; }
            const bool ray = w[p->obs_off + k * IMX_REC_WORDS + IMX_R_OP] == IMX_O_HEIGHT_SCAN;
            if (ray == (pass == 0)) order.push_back(c);
            if (ray && pass == 0) p->n_ray_cols++;
        }
    p->DC = (int)order.size();  // scheduled columns (= D without history and twins)
    const int DX = p->DC + (int)twins.size();
    {
        std::vector<int32_t> pad = order;
        pad.resize(p->D, 0);  // table keeps its D words
        w.insert(w.end(), pad.begin(), pad.end());
    }
    p->skip_off = (int)w.size();
    p->nskip = 0;

    // per-column expansion (see step.hip: XC_*), 16-byte aligned: the DC scheduled columns in `order`, then the twin columns.
    // XC_FLAGS bits 8..9 = group; a primary ray column's XC_AUX = 1 + index of the first entry of its twin chain, a twin's XC_AUX =
    // 1 + index of the next twin of the same ray (0 = end).
    while (w.size() % 4) w.push_back(0);
    p->xcol_off = (int)w.size();
    std::vector<int32_t> entry_cols = order;
    entry_cols.insert(entry_cols.end(), twins.begin(), twins.end());
    auto group_of_col = [&](int c) { int g = 0; while (g + 1 < p->ngroups && c >= p->gbase[g + 1]) ++g; return g; };
    std::vector<int> chain_head(p->DC, 0);
    std::vector<int32_t> xall((size_t)DX * 16, 0);
    for (int i = 0; i < DX; ++i) {
        const int c = entry_cols[i], k = col[c] >> 16, j = col[c] & 0xFFFF;
        const int g = group_of_col(c);
        const size_t ro = (size_t)p->obs_off + (size_t)k * IMX_REC_WORDS;
        int32_t* x = &xall[(size_t)i * 16];
        x[0] = c - p->gbase[g]; x[1] = w[ro + IMX_R_OP]; x[2] = j; x[3] = (w[ro + IMX_R_FLAGS] & 0xFF) | (g << 8) | (w[ro + IMX_R_FLAGS] & IMX_F_NOISE_GAUSS);
        x[4] = w[ro + IMX_R_P0]; x[5] = w[ro + IMX_R_NOISE_LO]; x[6] = w[ro + IMX_R_NOISE_HI];
        x[7] = w[ro + IMX_R_CLIP_LO]; x[8] = w[ro + IMX_R_CLIP_HI]; x[9] = w[ro + IMX_R_SCALE];
        const int op = x[1];
        const bool joint = op == IMX_O_JOINT_POS || op == IMX_O_JOINT_POS_REL || op == IMX_O_JOINT_VEL ||
                           op == IMX_O_JOINT_VEL_REL || op == IMX_O_JOINT_POS_LIMIT_NORMALIZED;
        if (joint) x[10] = w[(size_t)w[ro + IMX_R_IDS_OFF] + j];
        if (op == IMX_O_EXTERNAL) x[10] = w[ro + IMX_R_AUX0];
        if (op == IMX_O_HEIGHT_SCAN) {
            x[11] = w[(size_t)p->ray_off + 3 * j]; x[12] = w[(size_t)p->ray_off + 3 * j + 1]; x[13] = w[(size_t)p->ray_off + 3 * j + 2];
        }
        x[14] = (op != IMX_O_EXTERNAL && w[ro + IMX_R_AUX1] > 1) ? w[ro + IMX_R_AUX1] : 1;  // history slots
        x[15] = w[ro + IMX_R_DIM];                                                        // slot stride (term width)
    }
    for (int i = p->DC; i < DX; ++i) {  // link every twin behind the primary column of the same ray
        const int c = entry_cols[i], k = col[c] >> 16, j = col[c] & 0xFFFF;
        int head = -1;
        for (int q = 0; q < p->DC; ++q) {
            const int cq = entry_cols[q];
            if ((col[cq] >> 16) == twin_of[k] && (col[cq] & 0xFFFF) == j) { head = q; break; }
        }
        IMX_REQUIRE(head >= 0, "plan: twin height-scan column %d has no primary ray", c);
        int32_t* tail = &xall[(size_t)head * 16];
        while (tail[10] != 0) tail = &xall[(size_t)(tail[10] - 1) * 16];
        tail[10] = i + 1;
    }
    w.insert(w.end(), xall.begin(), xall.end());
    p->DX = DX;
    p->xmod_off = (int)w.size();
    p->MS = w[IMX_H_MOD_STATE];
    for (int i = 0; i < DX; ++i) {
        const int c = entry_cols[i], k = col[c] >> 16, j = col[c] & 0xFFFF;
        const size_t ro = (size_t)p->obs_off + (size_t)k * IMX_REC_WORDS;
        int32_t x[4] = {0, 0, 0, 0};
        if (w[ro + IMX_R_FLAGS] & IMX_F_MODIFIERS) {
            x[0] = w[ro + IMX_R_IDS2_OFF]; x[1] = w[ro + IMX_R_NIDS2]; x[2] = w[ro + IMX_R_P1] + j; x[3] = w[ro + IMX_R_DIM];
        }
        w.insert(w.end(), x, x + 4);
    }
    return 0;
}

static int upload_plan(imx_plan* p, hipStream_t stream, bool in_place) {
    if (imx_device_count() <= 0) return 0;
    auto& w = p->host;
    if (!in_place) {
        IMX_HIP(hipMalloc((void**)&p->dev, w.size() * sizeof(int32_t)));
        IMX_HIP(hipMemcpy(p->dev, w.data(), w.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    } else {
        IMX_HIP(hipMemcpyAsync(p->dev, w.data(), w.size() * sizeof(int32_t), hipMemcpyHostToDevice, stream));
    }
    return 0;
}

extern "C" int imx_plan_create(const int32_t* blob, size_t nwords, imx_plan_t** out) {
    IMX_REQUIRE(blob && out, "imx_plan_create: null argument");
    auto p = new imx_plan();
    std::unique_ptr<imx_plan> guard(p);
    if (parse_plan(blob, nwords, p)) return 1;
    if (upload_plan(p, nullptr, false)) return 1;
    *out = guard.release();
    return 0;
}

// RewardManager.set_term_cfg / TerminationManager.set_term_cfg (managers/reward_manager.py:163-176, termination_manager.py:207-220):
// a recompiled blob of the SAME shape replaces the tables in place -- the device pointer, and with it every hipGraph that captured
// a launch on this plan, stays valid; the copy is ordered on `stream` like the launches around it.
extern "C" int imx_plan_update(imx_plan_t* plan, const int32_t* blob, size_t nwords, imx_stream_t stream) {
    IMX_REQUIRE(plan && blob, "imx_plan_update: null argument");
    imx_plan q;
    if (parse_plan(blob, nwords, &q)) return 1;
    IMX_REQUIRE(q.host.size() == plan->host.size() && q.J == plan->J && q.B == plan->B && q.H == plan->H && q.A == plan->A &&
                    q.D == plan->D && q.R == plan->R && q.NB == plan->NB && q.nterm == plan->nterm && q.nrew_all == plan->nrew_all &&
                    q.nobs == plan->nobs && q.DC == plan->DC && q.DX == plan->DX && q.MS == plan->MS && q.ngroups == plan->ngroups &&
                    q.scan_stateful == plan->scan_stateful,
                "imx_plan_update: the new plan has a different shape (terms, widths or table sizes changed): create a new plan");
    for (int g = 0; g < q.ngroups; ++g) IMX_REQUIRE(q.gD[g] == plan->gD[g], "imx_plan_update: observation group %d changed width", g);
    int32_t* dev = plan->dev;
    *plan = q;  // host tables and offsets of the new blob
    plan->dev = dev;
    q.dev = nullptr;
    if (dev && upload_plan(plan, (hipStream_t)stream, true)) return 1;
    return 0;
}

extern "C" void imx_plan_destroy(imx_plan_t* plan) {
    if (!plan) return;
    if (plan->dev) (void)hipFree(plan->dev);
    delete plan;
}

extern "C" int imx_plan_obs_dim(const imx_plan_t* plan) { return plan ? plan->gD[0] : -1; }  // first group

PlanView imx_plan_view(const imx_plan* p) {
    PlanView v;
    const auto& w = p->host;
    v.w = p->dev;
    v.J = p->J; v.B = p->B; v.H = p->H; v.A = p->A; v.D = p->D; v.R = p->R; v.NB = p->NB; v.CMD = p->CMD;
    v.nterm = p->nterm; v.nrew = p->nrew; v.nobs = p->nobs; v.nact = p->nact; v.nrew_all = p->nrew_all;
    v.term_off = p->term_off; v.rew_off = p->rew_off; v.obs_off = p->obs_off; v.act_off = p->act_off;
    v.ray_off = p->ray_off; v.col_off = p->col_off; v.order_off = p->order_off; v.n_ray_cols = p->n_ray_cols;
    v.skip_off = p->skip_off; v.nskip = p->nskip; v.xcol_off = p->xcol_off; v.DC = p->DC; v.xmod_off = p->xmod_off; v.MS = p->MS;
    v.ngroups = p->ngroups;
    for (int g = 0; g < IMX_MAX_OBS_GROUPS; ++g) {
        v.gD[g] = g < p->ngroups ? p->gD[g] : 0;
        v.gbase[g] = g < p->ngroups ? p->gbase[g] : 0;
    }
    v.gcorrupt = 0;
    for (int g = 0; g < p->ngroups; ++g) v.gcorrupt |= p->gcorrupt[g] ? (1 << g) : 0;
    v.scan_stateful = p->scan_stateful ? 1 : 0;
    v.scan_substeps = w[IMX_H_SCAN_SUBSTEPS];
    v.scan_period = wf(w[IMX_H_SCAN_PERIOD]); v.scan_dt = wf(w[IMX_H_SCAN_DT]);
    v.drift_lo = wf(w[IMX_H_SCAN_DRIFT_LO]); v.drift_hi = wf(w[IMX_H_SCAN_DRIFT_HI]);
    v.max_ep_len = w[IMX_H_MAX_EP_LEN];
    v.step_dt = wf(w[IMX_H_STEP_DT]);
    v.max_ep_len_s = wf(w[IMX_H_MAX_EP_LEN_S]);
    v.gx = wf(w[IMX_H_GRAV_X]); v.gy = wf(w[IMX_H_GRAV_Y]); v.gz = wf(w[IMX_H_GRAV_Z]);
    v.rdx = wf(w[IMX_H_RAYDIR_X]); v.rdy = wf(w[IMX_H_RAYDIR_Y]); v.rdz = wf(w[IMX_H_RAYDIR_Z]);
    v.ray_max_dist = wf(w[IMX_H_RAY_MAXDIST]);
    v.rinv_dz = 1.0f / v.rdz;
    v.ray_yaw_only = w[IMX_H_RAY_YAW_ONLY];
    return v;
}

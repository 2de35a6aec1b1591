// aps_mex.cpp — the thin MATLAB mex gateway over libaps_hip.so's C ABI (include/aps.h).
//
// Build (on a machine that has MATLAB; NOT compiled in this repository's CI — there is no mex.h here):
//     mex -O -I../include aps_mex.cpp -L../<package>/lib -laps_hip
// One gateway, dispatched on a command string, replaces the reference's three mex files
// (PP/mex/flann_knn.cpp, nearest2HammingExhaustiveMEX.cpp, nearest2HammingExhaustiveOMPMEX.cpp) and adds
// the entry points the shadowing .m wrappers in this directory call.  Conventions kept from the reference
// gateways: inputs are borrowed column-major mxArrays (flann_knn.cpp:111-112), outputs are created with
// mxCreateNumericMatrix and owned by MATLAB (:193-194), indices are 1-based uint32, distances single, errors
// are raised with mexErrMsgIdAndTxt("aps:<kind>", ...) (cf. "flann_knn:type", "hamm2nn:cols").
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "aps.h"
#include "mex.h"

static void check(int rc) {
    if (rc == APS_OK) return;
    const char* kind = rc == APS_E_ARG ? "aps:args" : rc == APS_E_DIM ? "aps:dim" : rc == APS_E_TYPE ? "aps:type"
                     : rc == APS_E_OOM ? "aps:oom" : rc == APS_E_DEVICE ? "aps:device" : rc == APS_E_CAP ? "aps:cap" : "aps:internal";
    mexErrMsgIdAndTxt(kind, "%s", aps_last_error());
}
static std::string str(const mxArray* a) {
    char* c = mxArrayToString(a);
    std::string s(c ? c : "");
    if (c) mxFree(c);
    return s;
}
static double field(const mxArray* s, const char* name, double dflt) {
    const mxArray* f = mxIsStruct(s) ? mxGetField(s, 0, name) : nullptr;
    return f ? mxGetScalar(f) : dflt;
}
static void need(bool ok, const char* id, const char* msg) {
    if (!ok) mexErrMsgIdAndTxt(id, "%s", msg);
}

// [features, validPts] = aps_mex('sift_extract', img_uint8, input)
static void cmd_sift(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
    need(nrhs == 3 && mxIsUint8(prhs[1]), "aps:type", "usage: aps_mex('sift_extract', uint8 image, input struct)");
    const mwSize* d = mxGetDimensions(prhs[1]);
    const int h = (int)d[0], w = (int)d[1], c = mxGetNumberOfDimensions(prhs[1]) > 2 ? (int)d[2] : 1;
    aps_sift_params p;
    p.sigma = field(prhs[2], "Sigma", 1.6);
    p.n_layers = (int)field(prhs[2], "NumLayersInOctave", 4);
    p.contrast_threshold = field(prhs[2], "ContrastThreshold", 0.00133);
    p.edge_threshold = field(prhs[2], "EdgeThreshold", 6);
    p.max_features = 0;
    int64_t cap = (int64_t)h * w / 64 + 4096, count = 0;
    for (;;) {
        mxArray* desc = mxCreateNumericMatrix(cap, 128, mxSINGLE_CLASS, mxREAL);
        mxArray* loc = mxCreateNumericMatrix(cap, 2, mxDOUBLE_CLASS, mxREAL);
        const int rc = aps_sift_extract((const uint8_t*)mxGetData(prhs[1]), h, w, c, APS_IMG_U8_MATLAB, &p,
                                        (float*)mxGetData(desc), APS_COLMAJOR, cap, mxGetPr(loc), cap, nullptr, cap, &count);
        if (rc == APS_E_CAP && count > cap) {
            mxDestroyArray(desc);
            mxDestroyArray(loc);
            cap = count;
            continue;
        }
        check(rc);
        // shrink to count rows (column-major: compact each column)
        mxArray* f = mxCreateNumericMatrix(count, 128, mxSINGLE_CLASS, mxREAL);
        mxArray* v = mxCreateNumericMatrix(count, 2, mxDOUBLE_CLASS, mxREAL);
        for (int k = 0; k < 128; ++k)
            std::memcpy((float*)mxGetData(f) + (size_t)k * count, (float*)mxGetData(desc) + (size_t)k * cap, sizeof(float) * count);
        for (int k = 0; k < 2; ++k) std::memcpy(mxGetPr(v) + (size_t)k * count, mxGetPr(loc) + (size_t)k * cap, sizeof(double) * count);
        mxDestroyArray(desc);
        mxDestroyArray(loc);
        plhs[0] = f;
        if (nlhs > 1) plhs[1] = v; else mxDestroyArray(v);
        return;
    }
}

static aps_match_opts match_opts(const mxArray* s) {
    aps_match_opts o;
    o.max_ratio = field(s, "MaxRatio", 0.6);
    o.match_threshold = field(s, "MatchThreshold", 3.5);
    o.unique = field(s, "Unique", 1) != 0;
    o.normalize = 2;
    return o;
}

// [matches, metric] = aps_mex('match_features', F1 single N1x128, F2 single N2x128, opts)
static void cmd_match(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
    need(nrhs == 4 && mxIsSingle(prhs[1]) && mxIsSingle(prhs[2]), "aps:type", "descriptors must be single");
    const int64_t n1 = mxGetM(prhs[1]), n2 = mxGetM(prhs[2]);
    need(mxGetN(prhs[1]) == mxGetN(prhs[2]), "aps:dim", "Descriptor dimensions must match for non-binary.");
    const aps_match_opts o = match_opts(prhs[3]);
    std::vector<uint32_t> i1(n1 ? n1 : 1), i2(n1 ? n1 : 1);
    std::vector<float> met(n1 ? n1 : 1);
    int64_t k = 0;
    check(aps_match_features((const float*)mxGetData(prhs[1]), n1, n1, (const float*)mxGetData(prhs[2]), n2, n2,
                             (int)mxGetN(prhs[1]), APS_COLMAJOR, &o, i1.data(), i2.data(), met.data(), n1, &k));
    plhs[0] = mxCreateNumericMatrix(k, 2, mxUINT32_CLASS, mxREAL);
    uint32_t* m = (uint32_t*)mxGetData(plhs[0]);
    for (int64_t e = 0; e < k; ++e) { m[e] = i1[e]; m[e + k] = i2[e]; }
    if (nlhs > 1) {
        plhs[1] = mxCreateNumericMatrix(k, 1, mxSINGLE_CLASS, mxREAL);
        std::memcpy(mxGetData(plhs[1]), met.data(), sizeof(float) * k);
    }
}

// [idx2, dBest, dSecond] = aps_mex('pca_2nn', A single N1x128, B single N2x128, ApproxNumComponents, UsePCA)
// nearest2ApproxFloatFast (matchFeaturesScratch.m:442-573): idx2 uint32 N1x1 (1-based), distances single N1x1
static void cmd_pca2nn(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
    need(nrhs == 5 && mxIsSingle(prhs[1]) && mxIsSingle(prhs[2]), "aps:type", "descriptors must be single");
    const int64_t n1 = mxGetM(prhs[1]), n2 = mxGetM(prhs[2]);
    need(mxGetN(prhs[1]) == mxGetN(prhs[2]), "aps:dim", "Descriptor dimensions must match for non-binary.");
    need(n1 > 0 && n2 > 0, "aps:args", "Expected input to be nonempty.");
    plhs[0] = mxCreateNumericMatrix(n1, 1, mxUINT32_CLASS, mxREAL);
    mxArray* d1 = mxCreateNumericMatrix(n1, 1, mxSINGLE_CLASS, mxREAL);
    mxArray* d2 = mxCreateNumericMatrix(n1, 1, mxSINGLE_CLASS, mxREAL);
    check(aps_match_pca2nn((const float*)mxGetData(prhs[1]), n1, n1, (const float*)mxGetData(prhs[2]), n2, n2, (int)mxGetN(prhs[1]),
                           APS_COLMAJOR, (int)mxGetScalar(prhs[3]), mxGetScalar(prhs[4]) != 0, (uint32_t*)mxGetData(plhs[0]),
                           (float*)mxGetData(d1), (float*)mxGetData(d2), nullptr, nullptr));
    if (nlhs > 1) plhs[1] = d1; else mxDestroyArray(d1);
    if (nlhs > 2) plhs[2] = d2; else mxDestroyArray(d2);
}

// matches = aps_mex('match_pairwise', allDescriptors (1xN cell of single Ki x 128), opts) -> N x N cell, upper triangle
static void cmd_pairwise(int, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
    need(nrhs == 3 && mxIsCell(prhs[1]), "aps:type", "allDescriptors must be a cell array");
    const int n = (int)mxGetNumberOfElements(prhs[1]);
    std::vector<const float*> ptr(n);
    std::vector<int64_t> cnt(n), ld(n);
    int64_t cap = 0;
    for (int i = 0; i < n; ++i) {
        const mxArray* d = mxGetCell(prhs[1], i);
        need(d && mxIsSingle(d) && (mxGetM(d) == 0 || mxGetN(d) == 128), "aps:type", "descriptors must be single K x 128");
        ptr[i] = (const float*)mxGetData(d);
        cnt[i] = ld[i] = (int64_t)mxGetM(d);
        cap += cnt[i];
    }
    const int64_t np = (int64_t)n * (n - 1) / 2;
    const aps_match_opts o = match_opts(prhs[2]);
    std::vector<int64_t> pp(np + 1);
    std::vector<uint32_t> ii, jj;
    std::vector<float> met;
    int64_t k = 0;
    cap = cap / 4 + 1;
    for (;;) {
        ii.resize(cap); jj.resize(cap); met.resize(cap);
        const int rc = aps_match_pairwise(ptr.data(), cnt.data(), ld.data(), n, 128, APS_COLMAJOR, &o, pp.data(), ii.data(), jj.data(), met.data(), cap, &k);
        if (rc == APS_E_CAP) { cap = k; continue; }
        check(rc);
        break;
    }
    plhs[0] = mxCreateCellMatrix(n, n);
    int64_t p = 0;
    for (int j = 1; j < n; ++j)
        for (int i = 0; i < j; ++i, ++p) {  // featureMatchingPairwise.m:48 order
            const int64_t s = pp[p], m = pp[p + 1] - pp[p];
            mxArray* c = mxCreateDoubleMatrix(m, 2, mxREAL);  // double(matches), :120
            double* out = mxGetPr(c);
            for (int64_t e = 0; e < m; ++e) { out[e] = ii[s + e]; out[e + m] = jj[s + e]; }
            mxSetCell(plhs[0], i + (mwSize)j * n, c);
        }
}

// matches = aps_mex('match_global', allDescriptors (1xN cell of single Ki x 128), ratioThr, k) -> N x N cell, upper triangle
// featureMatchingGlobal.m:69-161 for float descriptors in one device pass: pooling, row normalisation (:80-86), the
// screened exact k-NN of the pool against itself (aps_knn_global_screened: queries the filter provably drops are not
// searched) and the per-query filter (:123-161, aps_global_filter).  Same lists as flann_knn_win + the reference's loop.
static void cmd_match_global(int, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
    need(nrhs == 4 && mxIsCell(prhs[1]), "aps:type", "usage: allDescriptors (cell of single K x 128), ratioThr, k");
    const int n = (int)mxGetNumberOfElements(prhs[1]);
    const double ratio = mxGetScalar(prhs[2]);
    const int k = (int)mxGetScalar(prhs[3]);
    need(k >= 1 && k <= 4, "aps:args", "k in 1..4 is built on the device (inputs.m: input.k = 4)");
    std::vector<int64_t> off(n + 1, 0);
    for (int i = 0; i < n; ++i) {
        const mxArray* d = mxGetCell(prhs[1], i);
        need(d && (mxGetM(d) == 0 || (mxIsSingle(d) && mxGetN(d) == 128)), "aps:type", "descriptors must be single K x 128");
        off[i + 1] = off[i] + (int64_t)mxGetM(d);
    }
    const int64_t f = off[n];
    plhs[0] = mxCreateCellMatrix(n, n);
    if (f == 0) return;
    // the pool, row-major (MATLAB's matrices are column-major K x 128)
    std::vector<float> raw((size_t)f * 128), pool((size_t)f * 128);
    std::vector<uint32_t> img((size_t)f), loc((size_t)f);
    for (int i = 0; i < n; ++i) {
        const mxArray* d = mxGetCell(prhs[1], i);
        const int64_t m = off[i + 1] - off[i];
        const float* src = m ? (const float*)mxGetData(d) : nullptr;
        for (int64_t r = 0; r < m; ++r) {
            for (int c = 0; c < 128; ++c) raw[(size_t)(off[i] + r) * 128 + c] = src[(size_t)c * m + r];
            img[(size_t)(off[i] + r)] = (uint32_t)(i + 1);
            loc[(size_t)(off[i] + r)] = (uint32_t)(r + 1);
        }
    }
    check(aps_global_normalize(raw.data(), f, 128, 128, APS_ROWMAJOR, pool.data()));
    std::vector<uint32_t> nn((size_t)f * k), oi((size_t)f), oj((size_t)f);
    std::vector<float> nd((size_t)f * k);
    check(aps_knn_global_screened(pool.data(), f, 128, 128, APS_ROWMAJOR, off.data(), n, (float)ratio, k, nn.data(), nd.data(), k));
    const int64_t np = (int64_t)n * (n - 1) / 2;
    std::vector<int64_t> pp(np + 1);
    int64_t cnt = 0;
    check(aps_global_filter(nn.data(), nd.data(), f, k, k, APS_ROWMAJOR, img.data(), loc.data(), n, (float)ratio, pp.data(), oi.data(),
                            oj.data(), f, &cnt));
    int64_t p = 0;
    for (int j = 1; j < n; ++j)
        for (int i = 0; i < j; ++i, ++p) {
            const int64_t s = pp[p], m = pp[p + 1] - pp[p];
            if (m == 0) continue;  // (the reference leaves such cells empty)
            mxArray* c = mxCreateDoubleMatrix(m, 2, mxREAL);
            double* out = mxGetPr(c);
            for (int64_t e = 0; e < m; ++e) { out[e] = oi[s + e]; out[e + m] = oj[s + e]; }
            mxSetCell(plhs[0], i + (mwSize)j * n, c);
        }
}

// [idx, dist] = aps_mex('knn_global', train, query, k)            (flann_knn_win contract)
static void cmd_knn(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
    const bool flt = nrhs >= 4 && mxIsSingle(prhs[1]) && mxIsSingle(prhs[2]);
    const bool bin = nrhs >= 4 && mxIsUint8(prhs[1]) && mxIsUint8(prhs[2]);
    need(flt || bin, "flann_knn:type", "Descriptors must be single (float) or uint8 (binary)");
    need(mxIsDouble(prhs[3]) && mxGetNumberOfElements(prhs[3]) == 1, "flann_knn:type", "k must be a scalar double");
    const int k = (int)mxGetScalar(prhs[3]);
    need(k > 0, "flann_knn:k", "k must be > 0");
    need(mxGetN(prhs[1]) == mxGetN(prhs[2]), "flann_knn:dim", "query must have same descriptor dimension as train");
    const int64_t ft = mxGetM(prhs[1]), fq = mxGetM(prhs[2]);
    plhs[0] = mxCreateNumericMatrix(fq, k, mxUINT32_CLASS, mxREAL);
    mxArray* dist = mxCreateNumericMatrix(fq, k, mxSINGLE_CLASS, mxREAL);
    if (bin)  // flann_knn.cpp:199-223 ('bf') and :235-240 (LSH): one exact Hamming k-NN
        check(aps_knn_hamming((const uint8_t*)mxGetData(prhs[1]), ft, ft, (const uint8_t*)mxGetData(prhs[2]), fq, fq,
                              (int)mxGetN(prhs[1]), APS_COLMAJOR, k, (uint32_t*)mxGetData(plhs[0]), (float*)mxGetData(dist), fq));
    else
    check(aps_knn_global((const float*)mxGetData(prhs[1]), ft, ft, (const float*)mxGetData(prhs[2]), fq, fq, (int)mxGetN(prhs[1]),
                         APS_COLMAJOR, k, (uint32_t*)mxGetData(plhs[0]), (float*)mxGetData(dist), fq));
    if (nlhs > 1) plhs[1] = dist; else mxDestroyArray(dist);
}

// [idx2, d1, d2] = aps_mex('hamming_2nn', Abytes, Bbytes)
static void cmd_hamming(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
    need(nrhs == 3, "hamm2nn:nrhs", "Need Abytes,Bbytes");
    need(mxIsUint8(prhs[1]) && mxIsUint8(prhs[2]), "hamm2nn:type", "Inputs must be uint8.");
    need(mxGetNumberOfDimensions(prhs[1]) == 2 && mxGetNumberOfDimensions(prhs[2]) == 2, "hamm2nn:dim", "2D only.");
    need(mxGetN(prhs[1]) == mxGetN(prhs[2]), "hamm2nn:cols", "Byte width mismatch.");
    const int64_t n1 = mxGetM(prhs[1]), n2 = mxGetM(prhs[2]);
    plhs[0] = mxCreateNumericMatrix(n1, 1, mxUINT32_CLASS, mxREAL);
    mxArray* d1 = mxCreateNumericMatrix(n1, 1, mxSINGLE_CLASS, mxREAL);
    mxArray* d2 = mxCreateNumericMatrix(n1, 1, mxSINGLE_CLASS, mxREAL);
    check(aps_hamming_2nn((const uint8_t*)mxGetData(prhs[1]), n1, n1, (const uint8_t*)mxGetData(prhs[2]), n2, n2, (int)mxGetN(prhs[1]),
                          APS_COLMAJOR, (uint32_t*)mxGetData(plhs[0]), (float*)mxGetData(d1), (float*)mxGetData(d2)));
    if (nlhs > 1) plhs[1] = d1; else mxDestroyArray(d1);
    if (nlhs > 2) plhs[2] = d2; else mxDestroyArray(d2);
}

// [model, inliers, isFound] = aps_mex('ransac_homography' | 'mlesac_homography', p1 Mx2, p2 Mx2, input, sampleIdx uint32 4xS
//                                      [, transformType = 'projective'])
static void cmd_ransac(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[], bool mlesac) {
    need((nrhs == 5 || nrhs == 6) && mxIsDouble(prhs[1]) && mxIsDouble(prhs[2]) && mxIsUint32(prhs[4]), "aps:type", "usage: p1, p2 double Mx2; sampleIdx uint32 4xS");
    need(mxGetM(prhs[4]) == 4, "aps:dim", "sampleIdx must have 4 rows");
    int tform = APS_TFORM_PROJECTIVE, min_pts = 4;
    if (nrhs == 6) {
        const std::string t = str(prhs[5]);
        if (t == "projective") tform = APS_TFORM_PROJECTIVE, min_pts = 4;
        else if (t == "affine") tform = APS_TFORM_AFFINE, min_pts = 3;
        else if (t == "similarity") tform = APS_TFORM_SIMILARITY, min_pts = 2;
        else if (t == "rigid") tform = APS_TFORM_RIGID, min_pts = 2;
        else if (t == "translation") tform = APS_TFORM_TRANSLATION, min_pts = 1;
        else mexErrMsgIdAndTxt("aps:type", "Unknown transform type");
    }
    const int64_t m = mxGetM(prhs[1]);
    need(mxGetM(prhs[2]) == (mwSize)m, "aps:dim", "matchedPoints1 and matchedPoints2 must have the same number of rows.");
    aps_ransac_opts o;
    o.max_distance = field(prhs[3], "maxDistance", 2.0);
    o.confidence = field(prhs[3], "inliersConfidence", 99.9);
    o.max_iter = (int)field(prhs[3], "maxIter", mlesac ? 1000 : 500);
    o.tform_type = tform;
    o.method = mlesac ? APS_ROBUST_MLESAC : APS_ROBUST_RANSAC;
    plhs[0] = mxCreateDoubleMatrix(3, 3, mxREAL);
    std::vector<uint8_t> mask(m ? m : 1);
    int found = 0;
    if (m >= min_pts)
        check(aps_ransac_homography(mxGetPr(prhs[1]), mxGetPr(prhs[2]), m, m, (const uint32_t*)mxGetData(prhs[4]), (int)mxGetN(prhs[4]), &o,
                                    mxGetPr(plhs[0]), mask.data(), &found, nullptr));
    if (nlhs > 1) {
        plhs[1] = mxCreateLogicalMatrix(m, 1);
        mxLogical* l = mxGetLogicals(plhs[1]);
        for (int64_t e = 0; e < m; ++e) l[e] = mask[e] != 0;
    }
    if (nlhs > 2) plhs[2] = mxCreateLogicalScalar(found != 0);
}

static void tform_of(const mxArray* a, int* tform, int* min_pts) {
    const std::string t = str(a);
    if (t == "projective") *tform = APS_TFORM_PROJECTIVE, *min_pts = 4;
    else if (t == "affine") *tform = APS_TFORM_AFFINE, *min_pts = 3;
    else if (t == "similarity") *tform = APS_TFORM_SIMILARITY, *min_pts = 2;
    else if (t == "rigid") *tform = APS_TFORM_RIGID, *min_pts = 2;
    else if (t == "translation") *tform = APS_TFORM_TRANSLATION, *min_pts = 1;
    else mexErrMsgIdAndTxt("aps:type", "Unknown transform type");
}

// sampleIdx = aps_mex('ransac_draw_samples', counts (P x 1 double: matches per pair), S, seed)
//   uint32 4 x S x P, column s of page p = one minimal-sample draw of pair p (1-based, local to the pair); the seeded
//   counter-based generator of aps_ransac_draw_samples, keyed by the pair's position (imageMatching.m:121 draws inside
//   parfor workers, whose streams the reference does not pin)
static void cmd_draw(int, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
    need(nrhs == 4 && mxIsDouble(prhs[1]), "aps:type", "usage: counts double, S, seed");
    const int P = (int)mxGetNumberOfElements(prhs[1]);
    const int S = (int)mxGetScalar(prhs[2]);
    need(S >= 1, "aps:args", "S must be positive");
    std::vector<int64_t> counts(P ? P : 1);
    std::vector<uint64_t> keys(P ? P : 1);
    for (int p = 0; p < P; ++p) {
        counts[p] = (int64_t)mxGetPr(prhs[1])[p];
        keys[p] = (uint64_t)p;
    }
    const mwSize d3[3] = {4, (mwSize)S, (mwSize)P};
    plhs[0] = mxCreateNumericArray(3, d3, mxUINT32_CLASS, mxREAL);
    if (P > 0)
        check(aps_ransac_draw_samples(counts.data(), keys.data(), P, S, (uint64_t)mxGetScalar(prhs[3]), (uint32_t*)mxGetData(plhs[0])));
}

// [models, mask, found, nInl] = aps_mex('ransac_homography_batch', pts1, pts2 (total x 2 double: the pairs' matched points one
//   after the other), pairPtr (P+1 double, 0-based row offsets), input struct, sampleIdx (uint32 4 x S x P), transformType,
//   method ('ransac' | 'mlesac'))  - every candidate pair of imageMatching.m:121-156 in one device batch
static void cmd_ransac_batch(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
    need(nrhs == 8 && mxIsDouble(prhs[1]) && mxIsDouble(prhs[2]) && mxIsDouble(prhs[3]) && mxIsUint32(prhs[5]), "aps:type",
         "usage: pts1, pts2 double total x 2; pairPtr double; input; sampleIdx uint32 4 x S x P; transformType; method");
    const int64_t total = (int64_t)mxGetM(prhs[1]);
    need(mxGetN(prhs[1]) == 2 && mxGetM(prhs[2]) == (mwSize)total && mxGetN(prhs[2]) == 2, "aps:dim", "pts1 and pts2 must both be total x 2");
    const int P = (int)mxGetNumberOfElements(prhs[3]) - 1;
    need(P >= 0, "aps:dim", "pairPtr needs P + 1 entries");
    std::vector<int64_t> pp(P + 1);
    for (int p = 0; p <= P; ++p) pp[p] = (int64_t)mxGetPr(prhs[3])[p];
    need(pp[0] == 0 && pp[P] == total, "aps:dim", "pairPtr must run from 0 to the number of rows");
    const mwSize* sd = mxGetDimensions(prhs[5]);
    const int nsd = (int)mxGetNumberOfDimensions(prhs[5]);
    need(sd[0] == 4 && (nsd > 2 ? (int)sd[2] : 1) == std::max(P, 1), "aps:dim", "sampleIdx must be 4 x S x P");
    aps_ransac_opts o;
    int min_pts = 4;
    tform_of(prhs[6], &o.tform_type, &min_pts);
    const bool mlesac = str(prhs[7]) == "mlesac";
    o.max_distance = field(prhs[4], "maxDistance", 2.0);
    o.confidence = field(prhs[4], "inliersConfidence", 99.9);
    o.max_iter = (int)field(prhs[4], "maxIter", mlesac ? 1000 : 500);
    o.method = mlesac ? APS_ROBUST_MLESAC : APS_ROBUST_RANSAC;
    const mwSize d3[3] = {3, 3, (mwSize)P};
    plhs[0] = mxCreateNumericArray(3, d3, mxDOUBLE_CLASS, mxREAL);
    std::vector<uint8_t> mask(total ? total : 1);
    std::vector<int32_t> found(P ? P : 1), ninl(P ? P : 1);
    if (P > 0 && total > 0)
        check(aps_ransac_homography_batch(mxGetPr(prhs[1]), mxGetPr(prhs[2]), total, pp.data(), P, (const uint32_t*)mxGetData(prhs[5]),
                                          (int)sd[1], &o, mxGetPr(plhs[0]), mask.data(), found.data(), ninl.data()));
    if (nlhs > 1) {
        plhs[1] = mxCreateLogicalMatrix(total, 1);
        mxLogical* l = mxGetLogicals(plhs[1]);
        for (int64_t e = 0; e < total; ++e) l[e] = mask[e] != 0;
    }
    if (nlhs > 2) {
        plhs[2] = mxCreateLogicalMatrix(P, 1);
        for (int p = 0; p < P; ++p) mxGetLogicals(plhs[2])[p] = found[p] != 0;
    }
    if (nlhs > 3) {
        plhs[3] = mxCreateDoubleMatrix(P, 1, mxREAL);
        for (int p = 0; p < P; ++p) mxGetPr(plhs[3])[p] = (double)ninl[p];
    }
}

// F = aps_mex('multiband_blend', Ci (1xK cell of single h x w x 3), Wi (1xK cell of single h x w), levels, sigma)
// F = aps_mex('linear_blend', Ci, Wi)
static void cmd_blend(bool multiband, int, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
    need(nrhs >= 3 && mxIsCell(prhs[1]) && mxIsCell(prhs[2]), "aps:type", "Ci and Wi must be cell arrays");
    const int K = (int)mxGetNumberOfElements(prhs[1]);
    need(K >= 1 && K == (int)mxGetNumberOfElements(prhs[2]), "aps:dim", "numel(Ci) must equal numel(Wi) >= 1");
    const mwSize* d = mxGetDimensions(mxGetCell(prhs[1], 0));
    const int h = (int)d[0], w = (int)d[1];
    const size_t hw = (size_t)h * w;
    std::vector<float> C(hw * 3 * K), W(hw * K), F(hw * 3);
    for (int k = 0; k < K; ++k) {  // MATLAB planar column-major -> row-major interleaved
        const mxArray* c = mxGetCell(prhs[1], k);
        const mxArray* wv = mxGetCell(prhs[2], k);
        need(mxIsSingle(c) && mxIsSingle(wv), "aps:type", "layers must be single");
        const float* cp = (const float*)mxGetData(c);
        const float* wp = (const float*)mxGetData(wv);
        const int ch = mxGetNumberOfDimensions(c) > 2 ? (int)mxGetDimensions(c)[2] : 1;
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                for (int q = 0; q < 3; ++q) C[((size_t)k * hw + (size_t)y * w + x) * 3 + q] = cp[(size_t)(ch == 1 ? 0 : q) * hw + (size_t)x * h + y];
                W[(size_t)k * hw + (size_t)y * w + x] = wp[(size_t)x * h + y];
            }
    }
    if (multiband)
        check(aps_multiband_blend(C.data(), W.data(), K, h, w, (int)mxGetScalar(prhs[3]), nrhs > 4 ? (float)mxGetScalar(prhs[4]) : 1.0f, F.data()));
    else
        check(aps_linear_blend(C.data(), W.data(), K, h, w, F.data()));
    const mwSize dims[3] = {(mwSize)h, (mwSize)w, 3};
    plhs[0] = mxCreateNumericArray(3, dims, mxSINGLE_CLASS, mxREAL);
    float* o = (float*)mxGetData(plhs[0]);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x)
            for (int q = 0; q < 3; ++q) o[(size_t)q * hw + (size_t)x * h + y] = F[((size_t)y * w + x) * 3 + q];
}

// [panorama, covered] = aps_mex('render', images (1xN cell uint8), cameras (struct array K,R), canvas struct, opts struct, gains Nx3)
static void cmd_render(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
    need(nrhs == 6 && mxIsCell(prhs[1]) && mxIsStruct(prhs[2]), "aps:type", "usage: images cell, cameras struct, canvas, opts, gains");
    const int n = (int)mxGetNumberOfElements(prhs[1]);
    std::vector<aps_image> im(n);
    const double* g = mxGetPr(prhs[5]);
    for (int i = 0; i < n; ++i) {
        const mxArray* a = mxGetCell(prhs[1], i);
        need(mxIsUint8(a), "aps:type", "images must be uint8");
        const mwSize* d = mxGetDimensions(a);
        im[i].data = (const uint8_t*)mxGetData(a);
        im[i].height = (int)d[0];
        im[i].width = (int)d[1];
        im[i].channels = mxGetNumberOfDimensions(a) > 2 ? (int)d[2] : 1;
        im[i].layout = APS_IMG_U8_MATLAB;
        std::memcpy(im[i].K, mxGetPr(mxGetField(prhs[2], i, "K")), 9 * sizeof(double));
        std::memcpy(im[i].R, mxGetPr(mxGetField(prhs[2], i, "R")), 9 * sizeof(double));
        for (int c = 0; c < 3; ++c) im[i].gain[c] = (float)g[i + (size_t)c * n];
    }
    aps_canvas cv;
    cv.mode = (int)field(prhs[3], "mode", APS_PROJ_SPHERICAL);
    cv.height = (int)field(prhs[3], "H", 0);
    cv.width = (int)field(prhs[3], "W", 0);
    cv.f_pan = field(prhs[3], "fPan", 1);
    cv.origin0 = field(prhs[3], "origin0", 0);
    cv.origin1 = field(prhs[3], "origin1", 0);
    const mxArray* rr = mxGetField(prhs[3], 0, "Rref");
    for (int e = 0; e < 9; ++e) cv.R_ref[e] = rr ? mxGetPr(rr)[e] : (e % 4 == 0);
    aps_render_opts ro;
    ro.tile_h = (int)field(prhs[4], "tileH", 2048);
    ro.tile_w = (int)field(prhs[4], "tileW", 2048);
    ro.angle_power = (float)field(prhs[4], "anglePower", 1);
    ro.blending = (int)field(prhs[4], "blendingId", APS_BLEND_MULTIBAND);
    ro.pyr_levels = (int)field(prhs[4], "pyrLevels", 3);
    ro.pyr_sigma = (float)field(prhs[4], "pyrSigma", 1);
    ro.none_policy = (int)field(prhs[4], "nonePolicyId", APS_NONE_LAST);
    ro.canvas_white = (int)field(prhs[4], "canvasWhite", 0);
    const mwSize dims[3] = {(mwSize)cv.height, (mwSize)cv.width, 3};
    plhs[0] = mxCreateNumericArray(3, dims, mxUINT8_CLASS, mxREAL);
    mxArray* cov = mxCreateNumericMatrix(cv.height, cv.width, mxUINT8_CLASS, mxREAL);
    check(aps_render(im.data(), n, &cv, &ro, APS_IMG_U8_MATLAB, (uint8_t*)mxGetData(plhs[0]), (uint8_t*)mxGetData(cov)));
    if (nlhs > 1) plhs[1] = cov; else mxDestroyArray(cov);
}

// [Nij, sumCi, sumCj] = aps_mex('gain_overlap_stats', images cell, cameras struct, canvas struct, stride)
static void cmd_gain(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
    need(nrhs == 5 && mxIsCell(prhs[1]) && mxIsStruct(prhs[2]), "aps:type", "usage: images cell, cameras struct, canvas, stride");
    const int n = (int)mxGetNumberOfElements(prhs[1]);
    std::vector<aps_image> im(n);
    for (int i = 0; i < n; ++i) {
        const mxArray* a = mxGetCell(prhs[1], i);
        need(mxIsUint8(a), "aps:type", "images must be uint8");
        const mwSize* d = mxGetDimensions(a);
        im[i].data = (const uint8_t*)mxGetData(a);
        im[i].height = (int)d[0];
        im[i].width = (int)d[1];
        im[i].channels = mxGetNumberOfDimensions(a) > 2 ? (int)d[2] : 1;
        im[i].layout = APS_IMG_U8_MATLAB;
        std::memcpy(im[i].K, mxGetPr(mxGetField(prhs[2], i, "K")), 9 * sizeof(double));
        std::memcpy(im[i].R, mxGetPr(mxGetField(prhs[2], i, "R")), 9 * sizeof(double));
        for (int c = 0; c < 3; ++c) im[i].gain[c] = 1.0f;
    }
    aps_canvas cv;
    cv.mode = (int)field(prhs[3], "mode", APS_PROJ_SPHERICAL);
    cv.height = (int)field(prhs[3], "H", 0);
    cv.width = (int)field(prhs[3], "W", 0);
    cv.f_pan = field(prhs[3], "fPan", 1);
    cv.origin0 = field(prhs[3], "origin0", 0);
    cv.origin1 = field(prhs[3], "origin1", 0);
    const mxArray* rr = mxGetField(prhs[3], 0, "Rref");
    for (int e = 0; e < 9; ++e) cv.R_ref[e] = rr ? mxGetPr(rr)[e] : (e % 4 == 0);
    const mwSize d3[3] = {(mwSize)n, (mwSize)n, 3};
    plhs[0] = mxCreateDoubleMatrix(n, n, mxREAL);
    mxArray* si = mxCreateNumericArray(3, d3, mxDOUBLE_CLASS, mxREAL);
    mxArray* sj = mxCreateNumericArray(3, d3, mxDOUBLE_CLASS, mxREAL);
    check(aps_gain_overlap_stats(im.data(), n, &cv, (int)mxGetScalar(prhs[4]), mxGetPr(plhs[0]), mxGetPr(si), mxGetPr(sj)));
    if (nlhs > 1) plhs[1] = si; else mxDestroyArray(si);
    if (nlhs > 2) plhs[2] = sj; else mxDestroyArray(sj);
}

// [Nij, sumCi, sumCj] = aps_mex('gain_overlap_stats_warped', Iw (1xN cell single HxWx3), Ww (1xN cell single HxW), ds)
// (the accumulation of gainCompensationH.m:45-52,78-149)
static void cmd_gain_warped(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
    need(nrhs == 4 && mxIsCell(prhs[1]) && mxIsCell(prhs[2]), "aps:type", "usage: Iw cell, Ww cell, overlapDownsample");
    const int n = (int)mxGetNumberOfElements(prhs[1]);
    need(n >= 1 && (int)mxGetNumberOfElements(prhs[2]) == n, "aps:type", "Iw and Ww must have the same length");
    std::vector<const float*> iw(n), ww(n);
    const mwSize* d0 = mxGetDimensions(mxGetCell(prhs[1], 0));
    const int64_t h = (int64_t)d0[0], w = (int64_t)d0[1];
    const int ch = mxGetNumberOfDimensions(mxGetCell(prhs[1], 0)) > 2 ? (int)d0[2] : 1;
    for (int i = 0; i < n; ++i) {
        const mxArray* a = mxGetCell(prhs[1], i);
        const mxArray* b = mxGetCell(prhs[2], i);
        need(mxIsSingle(a) && mxIsSingle(b), "aps:type", "Iw / Ww must be single");
        const mwSize* da = mxGetDimensions(a);
        const mwSize* db = mxGetDimensions(b);
        need((int64_t)da[0] == h && (int64_t)da[1] == w && (int64_t)db[0] == h && (int64_t)db[1] == w, "aps:type", "canvas sizes differ");
        iw[i] = (const float*)mxGetData(a);
        ww[i] = (const float*)mxGetData(b);
    }
    const mwSize d3[3] = {(mwSize)n, (mwSize)n, 3};
    plhs[0] = mxCreateDoubleMatrix(n, n, mxREAL);
    mxArray* si = mxCreateNumericArray(3, d3, mxDOUBLE_CLASS, mxREAL);
    mxArray* sj = mxCreateNumericArray(3, d3, mxDOUBLE_CLASS, mxREAL);
    check(aps_gain_overlap_stats_warped(iw.data(), ww.data(), n, h, w, ch, APS_COLMAJOR, (int)mxGetScalar(prhs[3]), mxGetPr(plhs[0]),
                                        mxGetPr(si), mxGetPr(sj)));
    if (nlhs > 1) plhs[1] = si; else mxDestroyArray(si);
    if (nlhs > 2) plhs[2] = sj; else mxDestroyArray(sj);
}

// J = aps_mex('imresize_u8', I uint8 HxWxC, [oh ow], [scale_r scale_c], bicubic(0/1))
static void cmd_imresize(int, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
    need(nrhs == 5 && mxIsUint8(prhs[1]), "aps:type", "usage: I uint8, [oh ow], [scale_r scale_c], bicubic");
    const mwSize* d = mxGetDimensions(prhs[1]);
    const int h = (int)d[0], w = (int)d[1], c = mxGetNumberOfDimensions(prhs[1]) > 2 ? (int)d[2] : 1;
    const double* sz = mxGetPr(prhs[2]);
    const double* sc = mxGetPr(prhs[3]);
    const mwSize od[3] = {(mwSize)sz[0], (mwSize)sz[1], (mwSize)c};
    plhs[0] = mxCreateNumericArray(c > 1 ? 3 : 2, od, mxUINT8_CLASS, mxREAL);
    check(aps_imresize_u8((const uint8_t*)mxGetData(prhs[1]), h, w, c, APS_IMG_U8_MATLAB, (int)sz[0], (int)sz[1], sc[0], sc[1],
                          mxGetScalar(prhs[4]) != 0 ? APS_RESIZE_BICUBIC : APS_RESIZE_BILINEAR, (uint8_t*)mxGetData(plhs[0])));
}

// [rect, valid] = aps_mex('crop_rect', I uint8 HxWx3, canvasWhite(0/1), range) ; rect = [offsetx offsety cropW cropH] (1-based)
static void cmd_crop(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
    need(nrhs == 4 && mxIsUint8(prhs[1]) && mxGetNumberOfDimensions(prhs[1]) == 3, "aps:type",
         "usage: I uint8 HxWx3, canvasWhite, range");
    const mwSize* d = mxGetDimensions(prhs[1]);
    need(d[2] == 3, "aps:dim", "I must have three channels");
    int32_t rect[4], valid = 0;
    check(aps_crop_rect((const uint8_t*)mxGetData(prhs[1]), (int64_t)d[0], (int64_t)d[1], APS_IMG_U8_MATLAB,
                        mxGetScalar(prhs[2]) != 0, mxGetScalar(prhs[3]), rect, &valid));
    plhs[0] = mxCreateDoubleMatrix(1, 4, mxREAL);
    for (int e = 0; e < 4; ++e) mxGetPr(plhs[0])[e] = rect[e];
    if (nlhs > 1) plhs[1] = mxCreateLogicalScalar(valid != 0);
}

// out = aps_mex('ba_pair_blocks', Ui total x 2, Uj total x 2, pairPtr (P+1, 0-based), cams 12 x 4 x P, sigmaHuber, bothDirections)
// out: 59 x P double = Hii, Hjj, Hij (4x4 column-major), gi, gj, E, r2sum, rcnt per pair
static void cmd_ba(int, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
    need(nrhs == 7 && mxIsDouble(prhs[1]) && mxIsDouble(prhs[2]) && mxIsDouble(prhs[4]), "aps:type",
         "usage: Ui, Uj (double total x 2), pairPtr, cams (12 x 4 x P double), sigmaHuber, bothDirections");
    const int64_t total = (int64_t)mxGetM(prhs[1]);
    need(mxGetN(prhs[1]) == 2 && mxGetN(prhs[2]) == 2 && (int64_t)mxGetM(prhs[2]) == total, "aps:dim", "Ui, Uj must be total x 2");
    const size_t np1 = mxGetNumberOfElements(prhs[3]);
    need(np1 >= 1, "aps:dim", "pairPtr must have P+1 entries");
    const int P = (int)np1 - 1;
    need(mxGetNumberOfElements(prhs[4]) == (size_t)48 * P, "aps:dim", "cams must be 12 x 4 x P");
    std::vector<int64_t> pp(np1);
    const double* src = mxGetPr(prhs[3]);
    for (size_t e = 0; e < np1; ++e) pp[e] = (int64_t)src[e];
    plhs[0] = mxCreateDoubleMatrix(59, P, mxREAL);
    check(aps_ba_pair_blocks(mxGetPr(prhs[1]), mxGetPr(prhs[2]), total, pp.data(), P, mxGetPr(prhs[4]), mxGetScalar(prhs[5]),
                             mxGetScalar(prhs[6]) != 0, mxGetPr(plhs[0])));
}

// warped = aps_mex('image_warp', image (uint8 or single, h x w x c), H (3x3 double), [oh ow], x0, y0, sx, sy, fillValue
//                  [, method: 0 nearest | 1 bilinear (default) | 2 bicubic])
// imageWarp.m:39-264.  MATLAB arrays are planar column-major; the C ABI wants row-major interleaved.
template <class T>
static mxArray* warp_typed(const mxArray* img, const double* H, int oh, int ow, double x0, double y0, double sx, double sy,
                           double fill, int method, mxClassID cls) {
    const mwSize* d = mxGetDimensions(img);
    const int h = (int)d[0], w = (int)d[1], c = mxGetNumberOfDimensions(img) > 2 ? (int)d[2] : 1;
    const T* src = (const T*)mxGetData(img);
    std::vector<T> in((size_t)h * w * c), out((size_t)oh * ow * c);
    for (int q = 0; q < c; ++q)
        for (int x = 0; x < w; ++x)
            for (int y = 0; y < h; ++y) in[((size_t)y * w + x) * c + q] = src[(size_t)q * h * w + (size_t)x * h + y];
    if (sizeof(T) == 1)
        check(aps_image_warp_u8((const uint8_t*)in.data(), h, w, c, H, oh, ow, x0, y0, sx, sy, (uint8_t)fill, method, (uint8_t*)out.data()));
    else
        check(aps_image_warp_f32((const float*)in.data(), h, w, c, H, oh, ow, x0, y0, sx, sy, (float)fill, method, (float*)out.data()));
    const mwSize dims[3] = {(mwSize)oh, (mwSize)ow, (mwSize)c};
    mxArray* o = mxCreateNumericArray(c > 1 ? 3 : 2, dims, cls, mxREAL);
    T* dst = (T*)mxGetData(o);
    for (int q = 0; q < c; ++q)
        for (int x = 0; x < ow; ++x)
            for (int y = 0; y < oh; ++y) dst[(size_t)q * oh * ow + (size_t)x * oh + y] = out[((size_t)y * ow + x) * c + q];
    return o;
}
static void cmd_image_warp(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
    (void)nlhs;
    need((nrhs == 9 || nrhs == 10) && (mxIsUint8(prhs[1]) || mxIsSingle(prhs[1])) && mxIsDouble(prhs[2]) && mxGetNumberOfElements(prhs[2]) == 9,
         "aps:type", "usage: image (uint8|single), H 3x3 double, [oh ow], x0, y0, sx, sy, fillValue[, method]");
    const double* sz = mxGetPr(prhs[3]);
    const int oh = (int)sz[0], ow = (int)sz[1];
    const double x0 = mxGetScalar(prhs[4]), y0 = mxGetScalar(prhs[5]), sx = mxGetScalar(prhs[6]), sy = mxGetScalar(prhs[7]);
    const double fill = mxGetScalar(prhs[8]);
    const int method = nrhs == 10 ? (int)mxGetScalar(prhs[9]) : APS_WARP_BILINEAR;
    if (mxIsUint8(prhs[1]))
        plhs[0] = warp_typed<uint8_t>(prhs[1], mxGetPr(prhs[2]), oh, ow, x0, y0, sx, sy, fill, method, mxUINT8_CLASS);
    else
        plhs[0] = warp_typed<float>(prhs[1], mxGetPr(prhs[2]), oh, ow, x0, y0, sx, sy, fill, method, mxSINGLE_CLASS);
}

// [rect, didCrop] = aps_mex('crop_nonzero_bbox', panorama uint8 h x w x 3, canvasWhite)   rect = [r1 r2 c1 c2]
static void cmd_crop_bbox(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
    need(nrhs == 3 && mxIsUint8(prhs[1]) && mxGetNumberOfDimensions(prhs[1]) == 3, "aps:type", "usage: uint8 h x w x 3 panorama, canvasWhite");
    const mwSize* d = mxGetDimensions(prhs[1]);
    int64_t rect[4];
    int did = 0;
    check(aps_crop_nonzero_bbox((const uint8_t*)mxGetData(prhs[1]), (int64_t)d[0], (int64_t)d[1], APS_IMG_U8_MATLAB,
                                mxGetScalar(prhs[2]) != 0, rect, &did));
    plhs[0] = mxCreateDoubleMatrix(1, 4, mxREAL);
    for (int e = 0; e < 4; ++e) mxGetPr(plhs[0])[e] = (double)rect[e];
    if (nlhs > 1) plhs[1] = mxCreateLogicalScalar(did != 0);
}

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
    need(nrhs >= 1 && mxIsChar(prhs[0]), "aps:args", "usage: aps_mex(command, ...)");
    const std::string cmd = str(prhs[0]);
    if (cmd == "version") plhs[0] = mxCreateDoubleScalar(aps_version());
    else if (cmd == "set_device") check(aps_set_device((int)mxGetScalar(prhs[1])));
    else if (cmd == "set_thread_stream_priority") check(aps_set_thread_stream_priority((int)mxGetScalar(prhs[1])));
    else if (cmd == "sift_extract") cmd_sift(nlhs, plhs, nrhs, prhs);
    else if (cmd == "match_features") cmd_match(nlhs, plhs, nrhs, prhs);
    else if (cmd == "pca_2nn") cmd_pca2nn(nlhs, plhs, nrhs, prhs);
    else if (cmd == "match_pairwise") cmd_pairwise(nlhs, plhs, nrhs, prhs);
    else if (cmd == "knn_global") cmd_knn(nlhs, plhs, nrhs, prhs);
    else if (cmd == "match_global") cmd_match_global(nlhs, plhs, nrhs, prhs);
    else if (cmd == "hamming_2nn") cmd_hamming(nlhs, plhs, nrhs, prhs);
    else if (cmd == "ransac_homography") cmd_ransac(nlhs, plhs, nrhs, prhs, false);
    else if (cmd == "mlesac_homography") cmd_ransac(nlhs, plhs, nrhs, prhs, true);
    else if (cmd == "ransac_homography_batch") cmd_ransac_batch(nlhs, plhs, nrhs, prhs);
    else if (cmd == "ransac_draw_samples") cmd_draw(nlhs, plhs, nrhs, prhs);
    else if (cmd == "multiband_blend") cmd_blend(true, nlhs, plhs, nrhs, prhs);
    else if (cmd == "linear_blend") cmd_blend(false, nlhs, plhs, nrhs, prhs);
    else if (cmd == "render") cmd_render(nlhs, plhs, nrhs, prhs);
    else if (cmd == "gain_overlap_stats") cmd_gain(nlhs, plhs, nrhs, prhs);
    else if (cmd == "gain_overlap_stats_warped") cmd_gain_warped(nlhs, plhs, nrhs, prhs);
    else if (cmd == "imresize_u8") cmd_imresize(nlhs, plhs, nrhs, prhs);
    else if (cmd == "crop_rect") cmd_crop(nlhs, plhs, nrhs, prhs);
    else if (cmd == "ba_pair_blocks") cmd_ba(nlhs, plhs, nrhs, prhs);
    else if (cmd == "image_warp") cmd_image_warp(nlhs, plhs, nrhs, prhs);
    else if (cmd == "crop_nonzero_bbox") cmd_crop_bbox(nlhs, plhs, nrhs, prhs);
    else mexErrMsgIdAndTxt("aps:args", "unknown command '%s'", cmd.c_str());
}

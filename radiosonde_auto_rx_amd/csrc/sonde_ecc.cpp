// sonde_ecc.cpp — block codes of the reference's bch_ecc_mod.c behind include/sonde_ecc.h: one descriptor-driven codec for
// RS(255,231), RS(255,223) CCSDS, BCH(63,51) and RS(15,11).  Own implementation of the same algorithm (key equation by the
// extended Euclidean algorithm on (S * sigma mod x^2t, x^2t) down to degree t + e/2, Chien search in the order x = 1, 2, ..,
// Forney with the code's first root b) with the reference's acceptance tests and failure codes, so every word — repairable,
// unrepairable or miscorrected — leaves it exactly as it leaves rs_decode_ErrEra (bch_ecc_mod.c:877-960).
#include "../../include/sonde_ecc.h"
#include <cstring>
#include <vector>

namespace {

enum { MAXD = 255 };                                           // MAX_DEG + 1 coefficients (bch_ecc_mod.h:31)

struct Poly {
    uint8_t c[MAXD];
    Poly() { memset(c, 0, MAXD); }
    int deg() const { int n = MAXD - 1; while (c[n] == 0 && n > 0) n--; if (c[n] == 0) n--; return n; }     // poly_deg (:403): deg(0) = -1
};

}  // namespace

struct sonde_ecc {
    int code, N, t, R, K, b, p, ip, ord;
    uint8_t exp_a[512], log_a[256];
    Poly g;
    uint8_t mul(uint8_t x, uint8_t y) const { return (x && y) ? exp_a[(log_a[x] + log_a[y]) % (ord - 1)] : 0; }
    uint8_t inv(uint8_t x) const { return x ? exp_a[(ord - 1 - log_a[x]) % (ord - 1)] : 0; }       // GF_inv (:367)
    uint8_t eval(const Poly &q, uint8_t x) const {             // poly_eval (:378): the value of sum_{n < ord-1} q[n] x^n
        uint8_t y = 0;
        for (int n = ord - 2; n >= 0; n--) y = mul(y, x) ^ q.c[n];
        return y;
    }
    Poly pmul(const Poly &a, const Poly &bq) const {           // poly_mul (:469)
        Poly r; const int da = a.deg(), db = bq.deg();
        if (da + db > MAXD - 1) return r;
        for (int i = 0; i <= da; i++) for (int j = 0; j <= db; j++) r.c[i + j] ^= mul(a.c[i], bq.c[j]);
        return r;
    }
    // p = d q + r, deg r < deg q (poly_divmod :410-454); q != 0
    void divmod(const Poly &pp, const Poly &q, Poly &d, Poly &r) const {
        d = Poly(); r = Poly();
        int dp = pp.deg(); const int dq = q.deg();
        if (dq < 0) return;
        if (dq == 0) { if (dp >= 0) { const uint8_t cf = mul(pp.c[dp], inv(q.c[0])); for (int i = 0; i <= dp; i++) d.c[i] = mul(pp.c[i], cf); } return; }
        if (dp < dq) { if (dp >= 0) for (int i = 0; i <= dp; i++) r.c[i] = pp.c[i]; return; }
        const uint8_t qi = inv(q.c[dq]);
        r = pp;
        uint8_t cf = mul(pp.c[dp], qi);
        while (dp >= dq) {
            d.c[dp - dq] = cf;
            for (int i = 0; i <= dq; i++) r.c[dp - i] ^= mul(q.c[dq - i], cf);
            while (r.c[dp] == 0 && dp > 0) dp--;
            if (r.c[dp] == 0) dp--;
            if (dp >= 0) cf = mul(r.c[dp], qi);
        }
    }
    int syndromes(const uint8_t *cw, Poly &S) const {          // S_i = cw((alpha^p)^(b+i)) (:638)
        int any = 0;
        Poly w; memcpy(w.c, cw, N);
        for (int i = 0; i < 2 * t; i++) {
            const uint8_t a_i = exp_a[(p * (b + i)) % (ord - 1)];
            S.c[i] = eval(w, a_i);
            any |= S.c[i] != 0;
        }
        return any;
    }
    // S Lambda = Omega mod x^2t, stop at deg(remainder) < deg (polyGF_lfsr :547)
    void lfsr(int deg, int x2t, const Poly &S, Poly &Lambda, Poly &Omega) const {
        Poly r0 = S, r1, s0, s1;
        r1.c[x2t] = 1; s0.c[0] = 1;
        while (r1.deg() >= deg) {
            Poly quo, rem; divmod(r0, r1, quo, rem);
            r0 = r1; r1 = rem;
            Poly s2 = pmul(quo, s1);
            for (int i = 0; i < MAXD; i++) s2.c[i] ^= s0.c[i];
            s0 = s1; s1 = s2;
        }
        Omega = r1; Lambda = s1;
    }
    uint8_t forney(uint8_t x, const Poly &Omega, const Poly &Lam) const {      // (:596)
        Poly D;
        for (int i = 1; i <= Lam.deg(); i++) if (i % 2) D.c[i - 1] = Lam.c[i];
        const uint8_t w = eval(Omega, x), z = eval(D, x);
        if (z == 0) return 0;
        uint8_t Y = mul(w, inv(z));
        if (b == 0) Y = mul(inv(x), Y);
        else if (b > 1) Y = mul(exp_a[((b - 1) * log_a[x]) % (ord - 1)], Y);
        return Y;
    }
};

static void gen_tables(sonde_ecc *c, unsigned f, int ord) {   // GF_genTab (:136): alpha = 2
    c->ord = ord;
    unsigned x = 1;
    for (int i = 0; i < ord - 1; i++) { c->exp_a[i] = (uint8_t)x; c->log_a[x] = (uint8_t)i; x <<= 1; if (x & (unsigned)ord) x ^= f; }
    for (int i = ord - 1; i < 512; i++) c->exp_a[i] = c->exp_a[i % (ord - 1)];
    c->log_a[0] = 0;
}

extern "C" {

sonde_ecc_t *sonde_ecc_create(int code) {
    sonde_ecc *c = new sonde_ecc();
    c->code = code; c->ip = 1;
    switch (code) {
        case SONDE_ECC_RS255:      c->N = 255; c->t = 12; c->R = 24; c->K = 231; c->b = 0;   c->p = 1;  gen_tables(c, 0x11D, 256); break;
        case SONDE_ECC_RS255CCSDS: c->N = 255; c->t = 16; c->R = 32; c->K = 223; c->b = 112; c->p = 11; gen_tables(c, 0x187, 256); break;
        case SONDE_ECC_BCH64:      c->N = 63;  c->t = 2;  c->R = 12; c->K = 51;  c->b = 1;   c->p = 1;  gen_tables(c, 0x43, 64); break;
        case SONDE_ECC_RS15CCSDS:  c->N = 15;  c->t = 2;  c->R = 4;  c->K = 11;  c->b = 6;   c->p = 1;  gen_tables(c, 0x13, 16); break;
        default: delete c; return nullptr;
    }
    for (int i = 1; i < c->ord - 1; i++) if ((c->p * i) % (c->ord - 1) == 1) { c->ip = i; break; }        // beta = alpha^p, beta^ip = alpha
    if (code == SONDE_ECC_BCH64) {                            // g = X^12+X^10+X^8+X^5+X^4+X^3+1 (rs_init_BCH64 :817)
        for (int k : {0, 3, 4, 5, 8, 10, 12}) c->g.c[k] = 1;
    } else {                                                  // g = prod (X - (alpha^p)^(b+i)), i < 2t
        c->g.c[0] = 1;
        for (int i = 0; i < 2 * c->t; i++) {
            Poly f; f.c[1] = 1; f.c[0] = c->exp_a[(c->p * (c->b + i)) % (c->ord - 1)];
            c->g = c->pmul(c->g, f);
        }
    }
    return c;
}

void sonde_ecc_destroy(sonde_ecc_t *c) { delete c; }

int sonde_ecc_params(const sonde_ecc_t *c, int *N, int *t, int *R, int *K) {
    if (!c) return -1;
    if (N) *N = c->N; if (t) *t = c->t; if (R) *R = c->R; if (K) *K = c->K;
    return 0;
}

int sonde_ecc_encode(const sonde_ecc_t *c, uint8_t *cw) {     // parity = message(X) mod g(X) (rs_encode :860)
    if (!c || !cw) return -1;
    Poly m, d, r;
    for (int j = c->R; j < c->N; j++) m.c[j] = cw[j];
    c->divmod(m, c->g, d, r);
    for (int j = 0; j < c->R; j++) cw[j] = r.c[j];
    return 0;
}

int sonde_ecc_decode_errera(const sonde_ecc_t *c, uint8_t *cw, int nera, const uint8_t *era_pos, uint8_t *err_pos, uint8_t *err_val) {
    if (!c || !cw || !err_pos || !err_val || (nera > 0 && !era_pos)) return -5;
    if (nera > 2 * c->t) return -4;
    for (int i = 0; i < 2 * c->t; i++) { err_pos[i] = 0; err_val[i] = 0; }
    Poly S;
    int errera = c->syndromes(cw, S);
    Poly sigma; sigma.c[0] = 1;
    if (nera > 0) {                                           // sigma = prod (1 - alpha^(p j) X) over the erasures (era_sigma :614)
        for (int i = 0; i < nera; i++) {
            Poly f; f.c[0] = 1; f.c[1] = c->exp_a[(c->p * era_pos[i]) % (c->ord - 1)];
            sigma = c->pmul(sigma, f);
        }
        S = c->pmul(sigma, S);
        for (int i = 2 * c->t; i < MAXD; i++) S.c[i] = 0;
    }
    if (!errera) return 0;
    Poly Lambda, Omega;
    c->lfsr(c->t + nera / 2, 2 * c->t, S, Lambda, Omega);
    const int dL = Lambda.deg(), dO = Omega.deg();
    if (dO >= dL + nera) return -3;
    const uint8_t gamma = Lambda.c[0];
    if (!gamma) return -2;
    const uint8_t gi = c->inv(gamma);
    for (int i = dL; i >= 0; i--) Lambda.c[i] = c->mul(Lambda.c[i], gi);
    for (int i = dO; i >= 0; i--) Omega.c[i] = c->mul(Omega.c[i], gi);
    const Poly sigLam = c->pmul(sigma, Lambda);
    const int dSL = sigLam.deg();
    int nerr = 0;
    for (int i = 1; i < c->ord; i++) {
        const uint8_t x = (uint8_t)i;
        if (c->eval(sigLam, x) == 0) {
            const uint8_t x1 = c->inv(x);
            err_pos[nerr] = (uint8_t)((c->log_a[x1] * c->ip) % (c->ord - 1));
            err_val[nerr] = c->forney(x, Omega, sigLam);
            nerr++;
        }
        if (nerr >= dSL) break;
    }
    if (nerr < dSL) return -1;
    for (int i = 0; i < nerr; i++) cw[err_pos[i]] ^= err_val[i];
    return nerr;
}

int sonde_ecc_decode(const sonde_ecc_t *c, uint8_t *cw, uint8_t *err_pos, uint8_t *err_val) {
    const uint8_t none[1] = {0};
    return sonde_ecc_decode_errera(c, cw, 0, none, err_pos, err_val);
}

int sonde_ecc_decode_bch_gf2t2(const sonde_ecc_t *c, uint8_t *cw, uint8_t *err_pos, uint8_t *err_val) {       // (:968)
    if (!c || !cw || !err_pos || !err_val) return -5;
    for (int i = 0; i < c->t; i++) { err_pos[i] = 0; err_val[i] = 0; }
    Poly S;
    int errors = c->syndromes(cw, S);
    if (!errors) return 0;
    Poly Lambda, Omega;
    c->lfsr(c->t, 2 * c->t, S, Lambda, Omega);
    const uint8_t gamma = Lambda.c[0];
    if (!gamma) return -2;
    const uint8_t gi = c->inv(gamma);
    for (int i = Lambda.deg(); i >= 0; i--) Lambda.c[i] = c->mul(Lambda.c[i], gi);
    for (int i = Omega.deg(); i >= 0; i--) Omega.c[i] = c->mul(Omega.c[i], gi);
    if (c->t == 2) {                                          // L(x) = 1 + S1 x + (S3 + S1^3)/S1 x^2 must be what Euclid found
        uint8_t L2 = c->mul(c->mul(S.c[0], S.c[0]), S.c[0]); L2 ^= S.c[2];
        L2 = c->mul(L2, c->inv(S.c[0]));
        if (S.c[1] != c->mul(S.c[0], S.c[0]) || S.c[3] != c->mul(S.c[1], S.c[1])) return -2;
        if (S.c[0] != Lambda.c[1] || L2 != Lambda.c[2]) return -2;
    }
    int n = 0;
    const int dL = Lambda.deg();
    for (int i = 1; i < c->ord; i++) {
        const uint8_t x = (uint8_t)i;
        if (c->eval(Lambda, x) == 0) { err_pos[n] = c->log_a[c->inv(x)]; err_val[n] = 1; n++; }
        if (n >= dL) break;
    }
    if (n < dL) return -1;
    for (int i = 0; i < n; i++) cw[err_pos[i]] ^= err_val[i];
    return n;
}

}  // extern "C"

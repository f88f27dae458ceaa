// sonde_gpsnav.cpp — see sonde_gpsnav.h.  Every floating-point expression keeps the reference's order of operations (nav_gps_vel.c), since
// the decoders print the results with five decimals and the tests compare the text.
#include "sonde_gpsnav.h"
#include <cmath>
#include <cstdlib>
#include <cstring>

namespace sonde {
namespace gpsnav {

namespace {

constexpr double kPi = 3.1415926535897932384626433832795;
constexpr double kRelClk = -4.442807633e-10;        // IS-GPS-200 F [s / sqrt(m)]
constexpr double kMu = 3.986005e14;                 // [m^3 / s^2]
constexpr double kOmegaE = 7.2921151467e-05;        // [rad / s]
constexpr double kWeek = 604800.0;
constexpr double kC = 299792458.0;
constexpr double kRangeEst = 0.072;                 // typical signal run time [s]: earth rotation between sending and receiving
constexpr double kEarthA = 6378137.0, kEarthB = 6356752.31424518;
constexpr double kA2B2 = kEarthA * kEarthA - kEarthB * kEarthB;

void rot_z(double x1, double y1, double z1, double angle, double *x2, double *y2, double *z2) {      // :59-65
    const double ca = cos(angle), sa = sin(angle);
    *x2 = ca * x1 + sa * y1;
    *y2 = -sa * x1 + ca * y1;
    *z2 = z1;
}

// 4x4 by cofactors, written over index sets: rows (a,b) x columns (c,d), rows r[3] x columns c[3] (:880-959, :1008-1080)
inline double det2(const double m[4][4], int a, int b, int c, int d) { return m[a][c] * m[b][d] - m[a][d] * m[b][c]; }
inline double det3(const double m[4][4], const int r[3], const int c[3]) {
    return m[r[0]][c[0]] * det2(m, r[1], r[2], c[1], c[2]) - m[r[0]][c[1]] * det2(m, r[1], r[2], c[0], c[2]) + m[r[0]][c[2]] * det2(m, r[1], r[2], c[0], c[1]);
}
const int kWithout[4][3] = { {1, 2, 3}, {0, 2, 3}, {0, 1, 3}, {0, 1, 2} };

double det4(const double m[4][4]) {
    return m[0][0] * det3(m, kWithout[0], kWithout[0]) - m[0][1] * det3(m, kWithout[0], kWithout[1])
         + m[0][2] * det3(m, kWithout[0], kWithout[2]) - m[0][3] * det3(m, kWithout[0], kWithout[3]);
}

int invert4(const double m[4][4], double inv[4][4]) {
    const double det = det4(m);
    if (fabs(det) < 0.0001) return -1;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            const double c = det3(m, kWithout[j], kWithout[i]);
            inv[i][j] = ((i + j) & 1) ? -c / det : c / det;
        }
    return 0;
}

int inverse_diagonal4(const double m[4][4], double diag[4]) {
    const double det = det4(m);
    if (fabs(det) < 0.0001) return -1;
    for (int i = 0; i < 4; i++) diag[i] = det3(m, kWithout[i], kWithout[i]) / det;
    return 0;
}

inline double lorentz(const double a[4], const double b[4]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] - a[3] * b[3]; }

// left inverse of the N x 4 matrix B: B^-1 for N = 4, (B^T B)^-1 B^T above (:1102-1123).  Where the 4x4 inverse fails (|det| < 1e-4) the reference goes on with an
// uninitialised matrix (its matrix_invert returns before writing); here that matrix is zero: no correction, no velocity.
void left_inverse(int N, const double B[][4], double Binv[4][12]) {
    double sq[4][4], sqinv[4][4];
    memset(sqinv, 0, sizeof sqinv);
    if (N == 4) {
        for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) sq[i][j] = B[i][j];
        invert4(sq, sqinv);
        for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) Binv[i][j] = sqinv[i][j];
        return;
    }
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            sq[i][j] = 0.0;
            for (int k = 0; k < N; k++) sq[i][j] += B[k][i] * B[k][j];
        }
    invert4(sq, sqinv);
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < N; j++) {
            Binv[i][j] = 0.0;
            for (int k = 0; k < 4; k++) Binv[i][j] += sqinv[i][k] * B[j][k];
        }
}

// one 19-character RINEX field ("D" exponents), value kept when the field does not parse — as the reference's sscanf leaves it
bool field19(FILE *fp, double *v) {
    char buf[24];
    if (fread(buf, 19, 1, fp) != 1) return false;
    if (buf[15] == 'D') buf[15] = 'E';
    buf[19] = 0;
    sscanf(buf, "%lf", v);
    return true;
}
bool skip3(FILE *fp) { char b[4]; return fread(b, 3, 1, fp) == 1; }
void rest_of_line(FILE *fp) { int c; while ((c = fgetc(fp)) != '\n') if (c == EOF) break; }

}  // namespace

void ecef2elli(double X, double Y, double Z, double *lat, double *lon, double *alt) {      // :29-51
    const double ea2 = kA2B2 / (kEarthA * kEarthA), eb2 = kA2B2 / (kEarthB * kEarthB);
    const double lam = atan2(Y, X);
    const double p = sqrt(X * X + Y * Y);
    const double t = atan2(Z * kEarthA, p * kEarthB);
    const double st = sin(t), ct = cos(t);
    const double phi = atan2(Z + eb2 * kEarthB * st * st * st, p - ea2 * kEarthA * ct * ct * ct);
    const double R = kEarthA / sqrt(1 - ea2 * sin(phi) * sin(phi));
    *alt = p / cos(phi) - R;
    *lat = phi * 180.0 / kPi;
    *lon = lam * 180.0 / kPi;
}

double dist3(double X1, double Y1, double Z1, double X2, double Y2, double Z2) {
    return sqrt((X2 - X1) * (X2 - X1) + (Y2 - Y1) * (Y2 - Y1) + (Z2 - Z1) * (Z2 - Z1));
}

int read_sem_almanac(FILE *fp, Eph alm[33]) {      // :132-179
    char name[64];
    unsigned n, week, toa, u;
    double v;
    if (fscanf(fp, "%u", &n) != 1) return -1;
    if (fscanf(fp, "%63s", name) != 1) return -1;
    if (fscanf(fp, "%u", &week) != 1) return -1;
    if (fscanf(fp, "%u", &toa) != 1) return -1;
    for (unsigned j = 1; j <= n && j < 33; j++) {
        Eph &a = alm[j];
        a.week = (uint16_t)week;
        a.toa = toa;
        a.toe = (double)toa;
        a.toc = a.toe;
        if (fscanf(fp, "%u", &u) != 1) return -1;   a.prn = (uint16_t)u;
        if (fscanf(fp, "%u", &u) != 1) return -2;   a.svn = (uint16_t)u;
        if (fscanf(fp, "%u", &u) != 1) return -3;   a.ura = (uint8_t)u;
        if (fscanf(fp, "%lf", &v) != 1) return -4;  a.e = v;
        if (fscanf(fp, "%lf", &v) != 1) return -5;  a.delta_i = v;  a.i0 = (0.30 + a.delta_i) * kPi;
        if (fscanf(fp, "%lf", &v) != 1) return -6;  a.OmegaDot = v * kPi;
        if (fscanf(fp, "%lf", &v) != 1) return -7;  a.sqrta = v;
        if (fscanf(fp, "%lf", &v) != 1) return -6;  a.Omega0 = v * kPi;
        if (fscanf(fp, "%lf", &v) != 1) return -8;  a.w = v * kPi;
        if (fscanf(fp, "%lf", &v) != 1) return -9;  a.M0 = v * kPi;
        if (fscanf(fp, "%lf", &v) != 1) return -10; a.af0 = v;
        if (fscanf(fp, "%lf", &v) != 1) return -11; a.af1 = v;
        a.af2 = a.crc = a.crs = a.cuc = a.cus = a.cic = a.cis = a.tgd = a.idot = a.delta_n = 0;
        if (fscanf(fp, "%u", &u) != 1) return -12;  a.health = (uint8_t)u;
        if (fscanf(fp, "%u", &u) != 1) return -13;  a.conf = (uint8_t)u;
    }
    return 0;
}

bool read_rinex_nav(FILE *fp, std::vector<Eph> &out) {      // read_RNXpephs :299-436: fixed columns, 8 lines per satellite and epoch
    char line[88];
    char *p;
    do {
        p = fgets(line, 84, fp);
        line[82] = '\0';
    } while (p && !strstr(line, "END OF HEADER"));
    if (p == NULL) return false;
    out.clear();
    Eph e;                     // fields of an entry that do not parse keep the previous entry's values
    double v = 0;
    unsigned prn = 0;
    for (;;) {
        char b[24];
        if (fread(b, 3, 1, fp) != 1) break;
        b[3] = 0;
        sscanf(b, "%d", (int *)&prn);
        e.prn = (uint16_t)prn;
        if (fread(b, 19, 1, fp) != 1) break;                    // epoch: not used
        if (!field19(fp, &v)) break; e.af0 = v;
        if (!field19(fp, &v)) break; e.af1 = v;
        if (!field19(fp, &v)) break; e.af2 = v;
        rest_of_line(fp);
        if (!skip3(fp)) break;
        if (!field19(fp, &v)) break;                            // iode
        if (!field19(fp, &v)) break; e.crs = v;
        if (!field19(fp, &v)) break; e.delta_n = v;
        if (!field19(fp, &v)) break; e.M0 = v;
        rest_of_line(fp);
        if (!skip3(fp)) break;
        if (!field19(fp, &v)) break; e.cuc = v;
        if (!field19(fp, &v)) break; e.e = v;
        if (!field19(fp, &v)) break; e.cus = v;
        if (!field19(fp, &v)) break; e.sqrta = v;
        rest_of_line(fp);
        if (!skip3(fp)) break;
        if (!field19(fp, &v)) break; e.toe = v; e.toc = e.toe;
        if (!field19(fp, &v)) break; e.cic = v;
        if (!field19(fp, &v)) break; e.Omega0 = v;
        if (!field19(fp, &v)) break; e.cis = v;
        rest_of_line(fp);
        if (!skip3(fp)) break;
        if (!field19(fp, &v)) break; e.i0 = v;
        if (!field19(fp, &v)) break; e.crc = v;
        if (!field19(fp, &v)) break; e.w = v;
        if (!field19(fp, &v)) break; e.OmegaDot = v;
        rest_of_line(fp);
        if (!skip3(fp)) break;
        if (!field19(fp, &v)) break; e.idot = v;
        if (!field19(fp, &v)) break;                            // codes on L2
        if (!field19(fp, &v)) break; e.gpsweek = (int)v;
        if (!field19(fp, &v)) break;                            // L2 P flag
        rest_of_line(fp);
        if (!skip3(fp)) break;
        if (!field19(fp, &v)) break;                            // accuracy
        if (!field19(fp, &v)) break; e.health = (uint8_t)(v + 0.1);
        if (!field19(fp, &v)) break; e.tgd = v;
        if (!field19(fp, &v)) break;                            // iodc
        rest_of_line(fp);
        if (!skip3(fp)) break;
        if (!field19(fp, &v)) break;                            // transmission time; the spare fields behind it may be missing
        p = fgets(line, 84, fp);
        e.week = 1;                                             // week numbers are taken relative to the entry (rollover -1 / 0 / +1 around toe)
        out.push_back(e);
        if (p == NULL) break;
    }
    Eph end;
    end.prn = 0;
    out.push_back(end);
    return true;
}

void sat_state(unsigned short week, double tow, const Eph &eph, bool with_velocity, Sat &s) {
    // clock (:444-502 / :1401-1463)
    const double a = eph.sqrta * eph.sqrta;
    double n = sqrt(kMu / (a * a * a));
    n += eph.delta_n;
    {
        const double tot = week * kWeek + tow;
        const double tk = tot - (eph.week * kWeek + eph.toe);
        const double tc = tot - (eph.week * kWeek + eph.toc);
        const double M = eph.M0 + n * tk;
        double E = M;
        for (int j = 0; j < 7; j++) E = M + eph.e * sin(E);
        double d_tr = kRelClk * eph.e * eph.sqrta * sin(E);
        d_tr *= kC;
        double d_tsv = eph.af0 + eph.af1 * tc + eph.af2 * tc * tc;
        d_tsv -= eph.tgd;
        s.clock_corr = d_tsv * kC + d_tr;
        if (with_velocity) s.clock_drift = (eph.af1 + 2.0 * eph.af2 * tc) * kC;
    }
    // the time the position is wanted for: corrected by the satellite clock (:654-664)
    unsigned short wk = week;
    double t = tow + s.clock_corr / kC;
    if (t < 0.0) { t += kWeek; wk--; }
    if (t > kWeek) { t -= kWeek; wk++; }

    // orbit (:558-625 / :1522-1633)
    const double tot = wk * kWeek + t;
    const double tk = tot - (eph.week * kWeek + eph.toe);
    const double M = eph.M0 + n * tk;
    double E = M;
    for (int j = 0; j < 7; j++) E = M + eph.e * sin(E);
    const double cosE = cos(E), sinE = sin(E);
    const double v = atan2(sqrt(1.0 - eph.e * eph.e) * sinE, cosE - eph.e);
    double u = v + eph.w;
    double r = a * (1.0 - eph.e * cos(E));
    double inc = eph.i0;
    double cos2u = cos(2.0 * u), sin2u = sin(2.0 * u);
    const double d_u = eph.cuc * cos2u + eph.cus * sin2u;
    const double d_r = eph.crc * cos2u + eph.crs * sin2u;
    const double d_i = eph.cic * cos2u + eph.cis * sin2u;
    u += d_u;
    r += d_r;
    inc += d_i + eph.idot * tk;
    const double cosu = cos(u), sinu = sin(u);
    const double x_op = r * cosu, y_op = r * sinu;
    const double omegak = eph.Omega0 + eph.OmegaDot * tk - kOmegaE * (tk + eph.toe);
    const double cos_ok = cos(omegak), sin_ok = sin(omegak);
    const double cosi = cos(inc), sini = sin(inc);
    s.X = x_op * cos_ok - y_op * sin_ok * cosi;
    s.Y = x_op * sin_ok + y_op * cos_ok * cosi;
    s.Z = y_op * sini;
    if (!with_velocity) return;

    // velocity: Remondi, GPS Solutions 8(3), 2004 (:1612-1633)
    cos2u = cos(2.0 * u);
    sin2u = sin(2.0 * u);
    const double edot = n / (1.0 - eph.e * cosE);
    const double vdot = sinE * edot * (1.0 + eph.e * cos(v)) / (sin(v) * (1.0 - eph.e * cosE));
    const double udot = vdot + 2.0 * (eph.cus * cos2u - eph.cuc * sin2u) * vdot;
    const double rdot = a * eph.e * sinE * n / (1.0 - eph.e * cosE) + 2.0 * (eph.crs * cos2u - eph.crc * sin2u) * vdot;
    const double idotdot = eph.idot + (eph.cis * cos2u - eph.cic * sin2u) * 2.0 * vdot;
    const double vx_op = rdot * cosu - y_op * udot;
    const double vy_op = rdot * sinu + x_op * udot;
    const double omegadotk = eph.OmegaDot - kOmegaE;
    const double ta = vx_op - y_op * cosi * omegadotk;
    const double tb = x_op * omegadotk + vy_op * cosi - y_op * sini * idotdot;
    s.vX = ta * cos_ok - tb * sin_ok;
    s.vY = ta * sin_ok + tb * cos_ok;
    s.vZ = vy_op * sini + y_op * cosi * idotdot;
}

int closed_form4(const Sat sats[4], double *latitude, double *longitude, double *height, double *rx_clock_bias, double pos_ecef[3]) {      // :682-874
    double p[4], x[4], y[4], z[4];
    for (int i = 0; i < 4; i++) p[i] = sats[i].pseudorange + sats[i].clock_corr;
    for (int i = 0; i < 4; i++) rot_z(sats[i].X, sats[i].Y, sats[i].Z, kOmegaE * kRangeEst, &x[i], &y[i], &z[i]);

    double dx[3], dy[3], dz[3], dp[3], k[3], A[3][3], D[3][3];
    for (int i = 0; i < 3; i++) {
        dx[i] = x[0] - x[i + 1]; dy[i] = y[0] - y[i + 1]; dz[i] = z[0] - z[i + 1];
        dp[i] = p[i + 1] - p[0];
    }
    for (int i = 0; i < 3; i++) {
        k[i] = dx[i] * dx[i] + dy[i] * dy[i] + dz[i] * dz[i] - dp[i] * dp[i];
        A[i][0] = 2.0 * dx[i]; A[i][1] = 2.0 * dy[i]; A[i][2] = 2.0 * dz[i];
    }
    double t1 = A[1][1] * A[2][2] - A[2][1] * A[1][2];
    double t2 = A[1][0] * A[2][2] - A[2][0] * A[1][2];
    double t3 = A[1][0] * A[2][1] - A[2][0] * A[1][1];
    const double detA = A[0][0] * t1 - A[0][1] * t2 + A[0][2] * t3;
    D[0][0] = t1;  D[1][0] = -t2;  D[2][0] = t3;
    D[0][1] = -A[0][1] * A[2][2] + A[2][1] * A[0][2];
    D[1][1] =  A[0][0] * A[2][2] - A[2][0] * A[0][2];
    D[2][1] = -A[0][0] * A[2][1] + A[2][0] * A[0][1];
    D[0][2] =  A[0][1] * A[1][2] - A[1][1] * A[0][2];
    D[1][2] = -A[0][0] * A[1][2] + A[1][0] * A[0][2];
    D[2][2] =  A[0][0] * A[1][1] - A[1][0] * A[0][1];

    double c[3], f[3];
    for (int i = 0; i < 3; i++) {
        c[i] = (D[i][0] * dp[0] + D[i][1] * dp[1] + D[i][2] * dp[2]) * 2.0 / detA;
        f[i] = (D[i][0] * k[0] + D[i][1] * k[1] + D[i][2] * k[2]) / detA;
    }
    const double m1 = c[0] * c[0] + c[1] * c[1] + c[2] * c[2] - 1.0;
    const double m2 = -2.0 * (c[0] * f[0] + c[1] * f[1] + c[2] * f[2]);
    const double m3 = f[0] * f[0] + f[1] * f[1] + f[2] * f[2];
    t1 = m2 * m2 - 4.0 * m1 * m3;
    if (t1 < 0) return -1;                                       // no real solution

    double d[2] = { (-m2 - sqrt(t1)) * 0.5 / m1, (-m2 + sqrt(t1)) * 0.5 / m1 };
    double s[2][3];
    for (int q = 0; q < 2; q++) {
        s[q][0] = c[0] * d[q] - f[0] + x[0];
        s[q][1] = c[1] * d[q] - f[1] + y[0];
        s[q][2] = c[2] * d[q] - f[2] + z[0];
    }
    // which of the two: the one near the earth's surface; when both are, the one nearer the point below the satellites' mean
    t1 = fabs(sqrt(s[0][0] * s[0][0] + s[0][1] * s[0][1] + s[0][2] * s[0][2]) - 6371000.0);
    t2 = fabs(sqrt(s[1][0] * s[1][0] + s[1][1] * s[1][1] + s[1][2] * s[1][2]) - 6371000.0);
    int pick = 0;
    double la, lo, al;
    if (t2 < t1 && t1 >= 60000) pick = 1;
    else if (t2 < 60000) {
        double laS, loS, alS, la2, lo2, al2;
        ecef2elli((x[0] + x[1] + x[2] + x[3]) / 4.0, (y[0] + y[1] + y[2] + y[3]) / 4.0, (z[0] + z[1] + z[2] + z[3]) / 4.0, &laS, &loS, &alS);
        ecef2elli(s[0][0], s[0][1], s[0][2], &la, &lo, &al);
        ecef2elli(s[1][0], s[1][1], s[1][2], &la2, &lo2, &al2);
        const double e1 = sqrt((laS - la) * (laS - la) + (loS - lo) * (loS - lo));
        const double e2 = sqrt((laS - la2) * (laS - la2) + (loS - lo2) * (loS - lo2));
        if (e2 < e1) pick = 1;
    }
    ecef2elli(s[pick][0], s[pick][1], s[pick][2], &la, &lo, &al);
    *latitude = la; *longitude = lo; *height = al;
    *rx_clock_bias = d[pick];
    for (int i = 0; i < 3; i++) pos_ecef[i] = s[pick][i];
    if (*height < -1500.0 || *height > 50000.0) return -2;
    return 0;
}

int dop(int n, const Sat *sats, const double pos_ecef[3], double DOP[4]) {      // calc_DOPn :961-991
    double G[12][4], GtG[4][4];
    if (n > 12) n = 12;
    for (int i = 0; i < n; i++) {
        const double d[3] = { sats[i].X - pos_ecef[0], sats[i].Y - pos_ecef[1], sats[i].Z - pos_ecef[2] };
        const double norm = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        for (int j = 0; j < 3; j++) G[i][j] = d[j] / norm;
        G[i][3] = 1;
    }
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            GtG[i][j] = 0.0;
            for (int k = 0; k < n; k++) GtG[i][j] += G[k][i] * G[k][j];
        }
    return inverse_diagonal4(GtG, DOP);
}

int bancroft(int N, const Sat *sats, double pos_ecef[3], double *cc) {      // NAV_bancroft1 :1082-1182
    double B[12][4], Binv[4][12], a[12], Be[4], Ba[4];
    if (N < 4 || N > 12) return -1;
    for (int i = 0; i < N; i++) {
        rot_z(sats[i].X, sats[i].Y, sats[i].Z, kOmegaE * kRangeEst, &B[i][0], &B[i][1], &B[i][2]);
        B[i][3] = sats[i].pseudorange + sats[i].clock_corr;
    }
    left_inverse(N, B, Binv);
    for (int i = 0; i < 4; i++) {
        Be[i] = 0.0;
        for (int k = 0; k < N; k++) Be[i] += Binv[i][k] * 1.0;
    }
    for (int i = 0; i < N; i++) a[i] = 0.5 * lorentz(B[i], B[i]);
    for (int i = 0; i < 4; i++) {
        Ba[i] = 0.0;
        for (int k = 0; k < N; k++) Ba[i] += Binv[i][k] * a[k];
    }
    const double q2 = lorentz(Be, Be), q1 = lorentz(Ba, Be) - 1, q0 = lorentz(Ba, Ba);
    if (q2 == 0) return -2;
    const double p = q1 / q2, q = q0 / q2;
    const double sq = p * p - q;
    if (sq < 0) return -2;
    const double root[2] = { -p + sqrt(sq), -p - sqrt(sq) };
    double L[2][4], off[2];
    for (int s = 0; s < 2; s++) {
        for (int i = 0; i < 4; i++) L[s][i] = root[s] * Be[i] + Ba[i];
        L[s][3] = -L[s][3];
        off[s] = fabs(sqrt(L[s][0] * L[s][0] + L[s][1] * L[s][1] + L[s][2] * L[s][2]) - 6371000.0);
    }
    const int pick = off[0] < off[1] ? 0 : 1;
    for (int i = 0; i < 3; i++) pos_ecef[i] = L[pick][i];
    *cc = L[pick][3];
    return 0;
}

int lin_pos(int N, const Sat *sats, const double pos_ecef[3], double dt, double dpos_ecef[3], double *cc) {      // NAV_LinP :1717-1796
    double B[12][4], Binv[4][12], a[12], norm[12], Ba[4];
    if (N < 4 || N > 12) return -1;
    for (int i = 0; i < N; i++) {
        double range = dist3(pos_ecef[0], pos_ecef[1], pos_ecef[2], sats[i].X, sats[i].Y, sats[i].Z);
        range /= kC;
        if (range < 0.06 || range > 0.1) range = kRangeEst;
        rot_z(sats[i].X, sats[i].Y, sats[i].Z, kOmegaE * range, &B[i][0], &B[i][1], &B[i][2]);
        const double X = B[i][0] - pos_ecef[0], Y = B[i][1] - pos_ecef[1], Z = B[i][2] - pos_ecef[2];
        norm[i] = sqrt(X * X + Y * Y + Z * Z);
        B[i][0] = X / norm[i]; B[i][1] = Y / norm[i]; B[i][2] = Z / norm[i];
        B[i][3] = 1;
    }
    left_inverse(N, B, Binv);
    for (int i = 0; i < N; i++) {
        const double obs_range = sats[i].pseudorange + sats[i].clock_corr;
        const double prox_range = norm[i] - dt;
        a[i] = prox_range - obs_range;
    }
    for (int i = 0; i < 4; i++) {
        Ba[i] = 0.0;
        for (int k = 0; k < N; k++) Ba[i] += Binv[i][k] * a[k];
    }
    for (int i = 0; i < 3; i++) dpos_ecef[i] = Ba[i];
    *cc = Ba[3];
    return 0;
}

int lin_vel(int N, const Sat *sats, const double pos_ecef[3], const double vel_ecef[3], double dt, double dvel_ecef[3], double *cc) {      // NAV_LinV :1798-1891
    double B[12][4], Binv[4][12], a[12], Ba[4];
    if (N < 4 || N > 12) return -1;
    for (int i = 0; i < N; i++) {
        rot_z(sats[i].X, sats[i].Y, sats[i].Z, kOmegaE * kRangeEst, &B[i][0], &B[i][1], &B[i][2]);
        const double X = B[i][0] - pos_ecef[0], Y = B[i][1] - pos_ecef[1], Z = B[i][2] - pos_ecef[2];
        const double norm = sqrt(X * X + Y * Y + Z * Z);
        B[i][0] = X / norm; B[i][1] = Y / norm; B[i][2] = Z / norm;
        B[i][3] = 1;
    }
    left_inverse(N, B, Binv);
    for (int i = 0; i < N; i++) {
        const double obs_rate = sats[i].pseudorate;
        // relative velocity projected on the line of sight, satellite position as sent (NAV_relVel :1699-1715)
        double x = sats[i].X - pos_ecef[0], y = sats[i].Y - pos_ecef[1], z = sats[i].Z - pos_ecef[2];
        const double norm = sqrt(x * x + y * y + z * z);
        x /= norm; y /= norm; z /= norm;
        const double v_proj = (sats[i].vX - vel_ecef[0]) * x + (sats[i].vY - vel_ecef[1]) * y + (sats[i].vZ - vel_ecef[2]) * z;
        const double prox_rate = v_proj - dt;
        a[i] = prox_rate - obs_rate;
    }
    for (int i = 0; i < 4; i++) {
        Ba[i] = 0.0;
        for (int k = 0; k < N; k++) Ba[i] += Binv[i][k] * a[k];
    }
    for (int i = 0; i < 3; i++) dvel_ecef[i] = Ba[i];
    *cc = Ba[3];
    return 0;
}

}  // namespace gpsnav
}  // namespace sonde

// sonde_gpsnav.h — GPS orbit data and single-epoch position / velocity solutions for sondes that send raw ranges (RS92).  Host code.
//
// Restates demod/mod/nav_gps_vel.c (IS-GPS-200 orbit model, closed-form 4-satellite solution, Bancroft, one linearised step, DOP) with the
// reference's order of operations, so that the printed coordinates agree to the last digit.
#pragma once
#include <cstdint>
#include <cstdio>
#include <vector>

namespace sonde {
namespace gpsnav {

struct Eph {                              // one broadcast ephemeris or almanac entry (EPHEM_t, nav_gps_vel.c:71-103)
    uint16_t prn = 0, week = 0;
    uint32_t toa = 0;
    double toe = 0, toc = 0, e = 0, delta_n = 0, delta_i = 0, i0 = 0, OmegaDot = 0, sqrta = 0, Omega0 = 0, w = 0, M0 = 0, tgd = 0, idot = 0;
    double cuc = 0, cus = 0, crc = 0, crs = 0, cic = 0, cis = 0, af0 = 0, af1 = 0, af2 = 0;
    int gpsweek = 0;
    uint16_t svn = 0;
    uint8_t ura = 0, health = 0, conf = 0;
};

struct Sat {                              // SAT_t :105-121 (the fields that are used)
    double pseudorange = 0, pseudorate = 0, clock_corr = 0, clock_drift = 0;
    double X = 0, Y = 0, Z = 0, vX = 0, vY = 0, vZ = 0;
    double PR = 0, ephtime = 0;
    int prn = 0;
};

int  read_sem_almanac(FILE *fp, Eph alm[33]);                       // 0 or the reference's negative codes
bool read_rinex_nav(FILE *fp, std::vector<Eph> &out);               // entries in file order, then one with prn 0

void ecef2elli(double X, double Y, double Z, double *lat, double *lon, double *alt);
double dist3(double X1, double Y1, double Z1, double X2, double Y2, double Z2);

// satellite clock correction [m] (+ drift [m/s]) and ECEF position (+ velocity) at gps week / time of week
void sat_state(unsigned short week, double tow, const Eph &e, bool with_velocity, Sat &s);

int  closed_form4(const Sat s[4], double *lat, double *lon, double *alt, double *rx_clock_bias, double pos_ecef[3]);
int  dop(int n, const Sat *s, const double pos_ecef[3], double DOP[4]);
int  bancroft(int N, const Sat *s, double pos_ecef[3], double *cc);
int  lin_pos(int N, const Sat *s, const double pos_ecef[3], double dt, double dpos_ecef[3], double *cc);
int  lin_vel(int N, const Sat *s, const double pos_ecef[3], const double vel_ecef[3], double dt, double dvel_ecef[3], double *cc);

}  // namespace gpsnav
}  // namespace sonde

// pfmi_fastmath.h -- domain-restricted fp64 transcendental kernels for the in-kernel normal generator.
//
// The Box-Muller transform only ever sees u = (x + 0.5) 2^-32 in (0,1), so the general-purpose ocml
// log / sqrt / sincospi (denormal, inf, NaN, huge-argument handling) are replaced by short sequences:
//   -2 ln u : exponent split + 128-entry table {inv_c, -log(inv_c)} + degree-6 log1p polynomial (|r| < 2^-8)
//   sqrt    : v_rsq_f64 seed + two Goldschmidt/Newton refinements (argument in [2e-10, 45])
//   sin/cos(2 pi u): exact quadrant reduction + Taylor polynomials on |pi f| <= pi/4
// Accuracy ~1e-15 absolute on the normals (checked against the ocml path by tools/microbench.hip and by
// the GPU parity tests, whose oracle uses glibc libm).
#pragma once
#include "pfmi_logtab.h"
#include "pfmi_sctab.h"

#ifdef __HIPCC__
__constant__ double PF_LOGTAB_DEV[128][2] = {
#define PF_ROW(i) {PF_LOGTAB_HOST[i][0], PF_LOGTAB_HOST[i][1]}
    PF_ROW(0), PF_ROW(1), PF_ROW(2), PF_ROW(3), PF_ROW(4), PF_ROW(5), PF_ROW(6), PF_ROW(7), PF_ROW(8), PF_ROW(9),
    PF_ROW(10), PF_ROW(11), PF_ROW(12), PF_ROW(13), PF_ROW(14), PF_ROW(15), PF_ROW(16), PF_ROW(17), PF_ROW(18), PF_ROW(19),
    PF_ROW(20), PF_ROW(21), PF_ROW(22), PF_ROW(23), PF_ROW(24), PF_ROW(25), PF_ROW(26), PF_ROW(27), PF_ROW(28), PF_ROW(29),
    PF_ROW(30), PF_ROW(31), PF_ROW(32), PF_ROW(33), PF_ROW(34), PF_ROW(35), PF_ROW(36), PF_ROW(37), PF_ROW(38), PF_ROW(39),
    PF_ROW(40), PF_ROW(41), PF_ROW(42), PF_ROW(43), PF_ROW(44), PF_ROW(45), PF_ROW(46), PF_ROW(47), PF_ROW(48), PF_ROW(49),
    PF_ROW(50), PF_ROW(51), PF_ROW(52), PF_ROW(53), PF_ROW(54), PF_ROW(55), PF_ROW(56), PF_ROW(57), PF_ROW(58), PF_ROW(59),
    PF_ROW(60), PF_ROW(61), PF_ROW(62), PF_ROW(63), PF_ROW(64), PF_ROW(65), PF_ROW(66), PF_ROW(67), PF_ROW(68), PF_ROW(69),
    PF_ROW(70), PF_ROW(71), PF_ROW(72), PF_ROW(73), PF_ROW(74), PF_ROW(75), PF_ROW(76), PF_ROW(77), PF_ROW(78), PF_ROW(79),
    PF_ROW(80), PF_ROW(81), PF_ROW(82), PF_ROW(83), PF_ROW(84), PF_ROW(85), PF_ROW(86), PF_ROW(87), PF_ROW(88), PF_ROW(89),
    PF_ROW(90), PF_ROW(91), PF_ROW(92), PF_ROW(93), PF_ROW(94), PF_ROW(95), PF_ROW(96), PF_ROW(97), PF_ROW(98), PF_ROW(99),
    PF_ROW(100), PF_ROW(101), PF_ROW(102), PF_ROW(103), PF_ROW(104), PF_ROW(105), PF_ROW(106), PF_ROW(107), PF_ROW(108), PF_ROW(109),
    PF_ROW(110), PF_ROW(111), PF_ROW(112), PF_ROW(113), PF_ROW(114), PF_ROW(115), PF_ROW(116), PF_ROW(117), PF_ROW(118), PF_ROW(119),
    PF_ROW(120), PF_ROW(121), PF_ROW(122), PF_ROW(123), PF_ROW(124), PF_ROW(125), PF_ROW(126), PF_ROW(127)
#undef PF_ROW
};


__constant__ double PF_SCTAB_DEV[256][2] = {
    {PF_SCTAB_HOST[0][0], PF_SCTAB_HOST[0][1]},
    {PF_SCTAB_HOST[1][0], PF_SCTAB_HOST[1][1]},
    {PF_SCTAB_HOST[2][0], PF_SCTAB_HOST[2][1]},
    {PF_SCTAB_HOST[3][0], PF_SCTAB_HOST[3][1]},
    {PF_SCTAB_HOST[4][0], PF_SCTAB_HOST[4][1]},
    {PF_SCTAB_HOST[5][0], PF_SCTAB_HOST[5][1]},
    {PF_SCTAB_HOST[6][0], PF_SCTAB_HOST[6][1]},
    {PF_SCTAB_HOST[7][0], PF_SCTAB_HOST[7][1]},
    {PF_SCTAB_HOST[8][0], PF_SCTAB_HOST[8][1]},
    {PF_SCTAB_HOST[9][0], PF_SCTAB_HOST[9][1]},
    {PF_SCTAB_HOST[10][0], PF_SCTAB_HOST[10][1]},
    {PF_SCTAB_HOST[11][0], PF_SCTAB_HOST[11][1]},
    {PF_SCTAB_HOST[12][0], PF_SCTAB_HOST[12][1]},
    {PF_SCTAB_HOST[13][0], PF_SCTAB_HOST[13][1]},
    {PF_SCTAB_HOST[14][0], PF_SCTAB_HOST[14][1]},
    {PF_SCTAB_HOST[15][0], PF_SCTAB_HOST[15][1]},
    {PF_SCTAB_HOST[16][0], PF_SCTAB_HOST[16][1]},
    {PF_SCTAB_HOST[17][0], PF_SCTAB_HOST[17][1]},
    {PF_SCTAB_HOST[18][0], PF_SCTAB_HOST[18][1]},
    {PF_SCTAB_HOST[19][0], PF_SCTAB_HOST[19][1]},
    {PF_SCTAB_HOST[20][0], PF_SCTAB_HOST[20][1]},
    {PF_SCTAB_HOST[21][0], PF_SCTAB_HOST[21][1]},
    {PF_SCTAB_HOST[22][0], PF_SCTAB_HOST[22][1]},
    {PF_SCTAB_HOST[23][0], PF_SCTAB_HOST[23][1]},
    {PF_SCTAB_HOST[24][0], PF_SCTAB_HOST[24][1]},
    {PF_SCTAB_HOST[25][0], PF_SCTAB_HOST[25][1]},
    {PF_SCTAB_HOST[26][0], PF_SCTAB_HOST[26][1]},
    {PF_SCTAB_HOST[27][0], PF_SCTAB_HOST[27][1]},
    {PF_SCTAB_HOST[28][0], PF_SCTAB_HOST[28][1]},
    {PF_SCTAB_HOST[29][0], PF_SCTAB_HOST[29][1]},
    {PF_SCTAB_HOST[30][0], PF_SCTAB_HOST[30][1]},
    {PF_SCTAB_HOST[31][0], PF_SCTAB_HOST[31][1]},
    {PF_SCTAB_HOST[32][0], PF_SCTAB_HOST[32][1]},
    {PF_SCTAB_HOST[33][0], PF_SCTAB_HOST[33][1]},
    {PF_SCTAB_HOST[34][0], PF_SCTAB_HOST[34][1]},
    {PF_SCTAB_HOST[35][0], PF_SCTAB_HOST[35][1]},
    {PF_SCTAB_HOST[36][0], PF_SCTAB_HOST[36][1]},
    {PF_SCTAB_HOST[37][0], PF_SCTAB_HOST[37][1]},
    {PF_SCTAB_HOST[38][0], PF_SCTAB_HOST[38][1]},
    {PF_SCTAB_HOST[39][0], PF_SCTAB_HOST[39][1]},
    {PF_SCTAB_HOST[40][0], PF_SCTAB_HOST[40][1]},
    {PF_SCTAB_HOST[41][0], PF_SCTAB_HOST[41][1]},
    {PF_SCTAB_HOST[42][0], PF_SCTAB_HOST[42][1]},
    {PF_SCTAB_HOST[43][0], PF_SCTAB_HOST[43][1]},
    {PF_SCTAB_HOST[44][0], PF_SCTAB_HOST[44][1]},
    {PF_SCTAB_HOST[45][0], PF_SCTAB_HOST[45][1]},
    {PF_SCTAB_HOST[46][0], PF_SCTAB_HOST[46][1]},
    {PF_SCTAB_HOST[47][0], PF_SCTAB_HOST[47][1]},
    {PF_SCTAB_HOST[48][0], PF_SCTAB_HOST[48][1]},
    {PF_SCTAB_HOST[49][0], PF_SCTAB_HOST[49][1]},
    {PF_SCTAB_HOST[50][0], PF_SCTAB_HOST[50][1]},
    {PF_SCTAB_HOST[51][0], PF_SCTAB_HOST[51][1]},
    {PF_SCTAB_HOST[52][0], PF_SCTAB_HOST[52][1]},
    {PF_SCTAB_HOST[53][0], PF_SCTAB_HOST[53][1]},
    {PF_SCTAB_HOST[54][0], PF_SCTAB_HOST[54][1]},
    {PF_SCTAB_HOST[55][0], PF_SCTAB_HOST[55][1]},
    {PF_SCTAB_HOST[56][0], PF_SCTAB_HOST[56][1]},
    {PF_SCTAB_HOST[57][0], PF_SCTAB_HOST[57][1]},
    {PF_SCTAB_HOST[58][0], PF_SCTAB_HOST[58][1]},
    {PF_SCTAB_HOST[59][0], PF_SCTAB_HOST[59][1]},
    {PF_SCTAB_HOST[60][0], PF_SCTAB_HOST[60][1]},
    {PF_SCTAB_HOST[61][0], PF_SCTAB_HOST[61][1]},
    {PF_SCTAB_HOST[62][0], PF_SCTAB_HOST[62][1]},
    {PF_SCTAB_HOST[63][0], PF_SCTAB_HOST[63][1]},
    {PF_SCTAB_HOST[64][0], PF_SCTAB_HOST[64][1]},
    {PF_SCTAB_HOST[65][0], PF_SCTAB_HOST[65][1]},
    {PF_SCTAB_HOST[66][0], PF_SCTAB_HOST[66][1]},
    {PF_SCTAB_HOST[67][0], PF_SCTAB_HOST[67][1]},
    {PF_SCTAB_HOST[68][0], PF_SCTAB_HOST[68][1]},
    {PF_SCTAB_HOST[69][0], PF_SCTAB_HOST[69][1]},
    {PF_SCTAB_HOST[70][0], PF_SCTAB_HOST[70][1]},
    {PF_SCTAB_HOST[71][0], PF_SCTAB_HOST[71][1]},
    {PF_SCTAB_HOST[72][0], PF_SCTAB_HOST[72][1]},
    {PF_SCTAB_HOST[73][0], PF_SCTAB_HOST[73][1]},
    {PF_SCTAB_HOST[74][0], PF_SCTAB_HOST[74][1]},
    {PF_SCTAB_HOST[75][0], PF_SCTAB_HOST[75][1]},
    {PF_SCTAB_HOST[76][0], PF_SCTAB_HOST[76][1]},
    {PF_SCTAB_HOST[77][0], PF_SCTAB_HOST[77][1]},
    {PF_SCTAB_HOST[78][0], PF_SCTAB_HOST[78][1]},
    {PF_SCTAB_HOST[79][0], PF_SCTAB_HOST[79][1]},
    {PF_SCTAB_HOST[80][0], PF_SCTAB_HOST[80][1]},
    {PF_SCTAB_HOST[81][0], PF_SCTAB_HOST[81][1]},
    {PF_SCTAB_HOST[82][0], PF_SCTAB_HOST[82][1]},
    {PF_SCTAB_HOST[83][0], PF_SCTAB_HOST[83][1]},
    {PF_SCTAB_HOST[84][0], PF_SCTAB_HOST[84][1]},
    {PF_SCTAB_HOST[85][0], PF_SCTAB_HOST[85][1]},
    {PF_SCTAB_HOST[86][0], PF_SCTAB_HOST[86][1]},
    {PF_SCTAB_HOST[87][0], PF_SCTAB_HOST[87][1]},
    {PF_SCTAB_HOST[88][0], PF_SCTAB_HOST[88][1]},
    {PF_SCTAB_HOST[89][0], PF_SCTAB_HOST[89][1]},
    {PF_SCTAB_HOST[90][0], PF_SCTAB_HOST[90][1]},
    {PF_SCTAB_HOST[91][0], PF_SCTAB_HOST[91][1]},
    {PF_SCTAB_HOST[92][0], PF_SCTAB_HOST[92][1]},
    {PF_SCTAB_HOST[93][0], PF_SCTAB_HOST[93][1]},
    {PF_SCTAB_HOST[94][0], PF_SCTAB_HOST[94][1]},
    {PF_SCTAB_HOST[95][0], PF_SCTAB_HOST[95][1]},
    {PF_SCTAB_HOST[96][0], PF_SCTAB_HOST[96][1]},
    {PF_SCTAB_HOST[97][0], PF_SCTAB_HOST[97][1]},
    {PF_SCTAB_HOST[98][0], PF_SCTAB_HOST[98][1]},
    {PF_SCTAB_HOST[99][0], PF_SCTAB_HOST[99][1]},
    {PF_SCTAB_HOST[100][0], PF_SCTAB_HOST[100][1]},
    {PF_SCTAB_HOST[101][0], PF_SCTAB_HOST[101][1]},
    {PF_SCTAB_HOST[102][0], PF_SCTAB_HOST[102][1]},
    {PF_SCTAB_HOST[103][0], PF_SCTAB_HOST[103][1]},
    {PF_SCTAB_HOST[104][0], PF_SCTAB_HOST[104][1]},
    {PF_SCTAB_HOST[105][0], PF_SCTAB_HOST[105][1]},
    {PF_SCTAB_HOST[106][0], PF_SCTAB_HOST[106][1]},
    {PF_SCTAB_HOST[107][0], PF_SCTAB_HOST[107][1]},
    {PF_SCTAB_HOST[108][0], PF_SCTAB_HOST[108][1]},
    {PF_SCTAB_HOST[109][0], PF_SCTAB_HOST[109][1]},
    {PF_SCTAB_HOST[110][0], PF_SCTAB_HOST[110][1]},
    {PF_SCTAB_HOST[111][0], PF_SCTAB_HOST[111][1]},
    {PF_SCTAB_HOST[112][0], PF_SCTAB_HOST[112][1]},
    {PF_SCTAB_HOST[113][0], PF_SCTAB_HOST[113][1]},
    {PF_SCTAB_HOST[114][0], PF_SCTAB_HOST[114][1]},
    {PF_SCTAB_HOST[115][0], PF_SCTAB_HOST[115][1]},
    {PF_SCTAB_HOST[116][0], PF_SCTAB_HOST[116][1]},
    {PF_SCTAB_HOST[117][0], PF_SCTAB_HOST[117][1]},
    {PF_SCTAB_HOST[118][0], PF_SCTAB_HOST[118][1]},
    {PF_SCTAB_HOST[119][0], PF_SCTAB_HOST[119][1]},
    {PF_SCTAB_HOST[120][0], PF_SCTAB_HOST[120][1]},
    {PF_SCTAB_HOST[121][0], PF_SCTAB_HOST[121][1]},
    {PF_SCTAB_HOST[122][0], PF_SCTAB_HOST[122][1]},
    {PF_SCTAB_HOST[123][0], PF_SCTAB_HOST[123][1]},
    {PF_SCTAB_HOST[124][0], PF_SCTAB_HOST[124][1]},
    {PF_SCTAB_HOST[125][0], PF_SCTAB_HOST[125][1]},
    {PF_SCTAB_HOST[126][0], PF_SCTAB_HOST[126][1]},
    {PF_SCTAB_HOST[127][0], PF_SCTAB_HOST[127][1]},
    {PF_SCTAB_HOST[128][0], PF_SCTAB_HOST[128][1]},
    {PF_SCTAB_HOST[129][0], PF_SCTAB_HOST[129][1]},
    {PF_SCTAB_HOST[130][0], PF_SCTAB_HOST[130][1]},
    {PF_SCTAB_HOST[131][0], PF_SCTAB_HOST[131][1]},
    {PF_SCTAB_HOST[132][0], PF_SCTAB_HOST[132][1]},
    {PF_SCTAB_HOST[133][0], PF_SCTAB_HOST[133][1]},
    {PF_SCTAB_HOST[134][0], PF_SCTAB_HOST[134][1]},
    {PF_SCTAB_HOST[135][0], PF_SCTAB_HOST[135][1]},
    {PF_SCTAB_HOST[136][0], PF_SCTAB_HOST[136][1]},
    {PF_SCTAB_HOST[137][0], PF_SCTAB_HOST[137][1]},
    {PF_SCTAB_HOST[138][0], PF_SCTAB_HOST[138][1]},
    {PF_SCTAB_HOST[139][0], PF_SCTAB_HOST[139][1]},
    {PF_SCTAB_HOST[140][0], PF_SCTAB_HOST[140][1]},
    {PF_SCTAB_HOST[141][0], PF_SCTAB_HOST[141][1]},
    {PF_SCTAB_HOST[142][0], PF_SCTAB_HOST[142][1]},
    {PF_SCTAB_HOST[143][0], PF_SCTAB_HOST[143][1]},
    {PF_SCTAB_HOST[144][0], PF_SCTAB_HOST[144][1]},
    {PF_SCTAB_HOST[145][0], PF_SCTAB_HOST[145][1]},
    {PF_SCTAB_HOST[146][0], PF_SCTAB_HOST[146][1]},
    {PF_SCTAB_HOST[147][0], PF_SCTAB_HOST[147][1]},
    {PF_SCTAB_HOST[148][0], PF_SCTAB_HOST[148][1]},
    {PF_SCTAB_HOST[149][0], PF_SCTAB_HOST[149][1]},
    {PF_SCTAB_HOST[150][0], PF_SCTAB_HOST[150][1]},
    {PF_SCTAB_HOST[151][0], PF_SCTAB_HOST[151][1]},
    {PF_SCTAB_HOST[152][0], PF_SCTAB_HOST[152][1]},
    {PF_SCTAB_HOST[153][0], PF_SCTAB_HOST[153][1]},
    {PF_SCTAB_HOST[154][0], PF_SCTAB_HOST[154][1]},
    {PF_SCTAB_HOST[155][0], PF_SCTAB_HOST[155][1]},
    {PF_SCTAB_HOST[156][0], PF_SCTAB_HOST[156][1]},
    {PF_SCTAB_HOST[157][0], PF_SCTAB_HOST[157][1]},
    {PF_SCTAB_HOST[158][0], PF_SCTAB_HOST[158][1]},
    {PF_SCTAB_HOST[159][0], PF_SCTAB_HOST[159][1]},
    {PF_SCTAB_HOST[160][0], PF_SCTAB_HOST[160][1]},
    {PF_SCTAB_HOST[161][0], PF_SCTAB_HOST[161][1]},
    {PF_SCTAB_HOST[162][0], PF_SCTAB_HOST[162][1]},
    {PF_SCTAB_HOST[163][0], PF_SCTAB_HOST[163][1]},
    {PF_SCTAB_HOST[164][0], PF_SCTAB_HOST[164][1]},
    {PF_SCTAB_HOST[165][0], PF_SCTAB_HOST[165][1]},
    {PF_SCTAB_HOST[166][0], PF_SCTAB_HOST[166][1]},
    {PF_SCTAB_HOST[167][0], PF_SCTAB_HOST[167][1]},
    {PF_SCTAB_HOST[168][0], PF_SCTAB_HOST[168][1]},
    {PF_SCTAB_HOST[169][0], PF_SCTAB_HOST[169][1]},
    {PF_SCTAB_HOST[170][0], PF_SCTAB_HOST[170][1]},
    {PF_SCTAB_HOST[171][0], PF_SCTAB_HOST[171][1]},
    {PF_SCTAB_HOST[172][0], PF_SCTAB_HOST[172][1]},
    {PF_SCTAB_HOST[173][0], PF_SCTAB_HOST[173][1]},
    {PF_SCTAB_HOST[174][0], PF_SCTAB_HOST[174][1]},
    {PF_SCTAB_HOST[175][0], PF_SCTAB_HOST[175][1]},
    {PF_SCTAB_HOST[176][0], PF_SCTAB_HOST[176][1]},
    {PF_SCTAB_HOST[177][0], PF_SCTAB_HOST[177][1]},
    {PF_SCTAB_HOST[178][0], PF_SCTAB_HOST[178][1]},
    {PF_SCTAB_HOST[179][0], PF_SCTAB_HOST[179][1]},
    {PF_SCTAB_HOST[180][0], PF_SCTAB_HOST[180][1]},
    {PF_SCTAB_HOST[181][0], PF_SCTAB_HOST[181][1]},
    {PF_SCTAB_HOST[182][0], PF_SCTAB_HOST[182][1]},
    {PF_SCTAB_HOST[183][0], PF_SCTAB_HOST[183][1]},
    {PF_SCTAB_HOST[184][0], PF_SCTAB_HOST[184][1]},
    {PF_SCTAB_HOST[185][0], PF_SCTAB_HOST[185][1]},
    {PF_SCTAB_HOST[186][0], PF_SCTAB_HOST[186][1]},
    {PF_SCTAB_HOST[187][0], PF_SCTAB_HOST[187][1]},
    {PF_SCTAB_HOST[188][0], PF_SCTAB_HOST[188][1]},
    {PF_SCTAB_HOST[189][0], PF_SCTAB_HOST[189][1]},
    {PF_SCTAB_HOST[190][0], PF_SCTAB_HOST[190][1]},
    {PF_SCTAB_HOST[191][0], PF_SCTAB_HOST[191][1]},
    {PF_SCTAB_HOST[192][0], PF_SCTAB_HOST[192][1]},
    {PF_SCTAB_HOST[193][0], PF_SCTAB_HOST[193][1]},
    {PF_SCTAB_HOST[194][0], PF_SCTAB_HOST[194][1]},
    {PF_SCTAB_HOST[195][0], PF_SCTAB_HOST[195][1]},
    {PF_SCTAB_HOST[196][0], PF_SCTAB_HOST[196][1]},
    {PF_SCTAB_HOST[197][0], PF_SCTAB_HOST[197][1]},
    {PF_SCTAB_HOST[198][0], PF_SCTAB_HOST[198][1]},
    {PF_SCTAB_HOST[199][0], PF_SCTAB_HOST[199][1]},
    {PF_SCTAB_HOST[200][0], PF_SCTAB_HOST[200][1]},
    {PF_SCTAB_HOST[201][0], PF_SCTAB_HOST[201][1]},
    {PF_SCTAB_HOST[202][0], PF_SCTAB_HOST[202][1]},
    {PF_SCTAB_HOST[203][0], PF_SCTAB_HOST[203][1]},
    {PF_SCTAB_HOST[204][0], PF_SCTAB_HOST[204][1]},
    {PF_SCTAB_HOST[205][0], PF_SCTAB_HOST[205][1]},
    {PF_SCTAB_HOST[206][0], PF_SCTAB_HOST[206][1]},
    {PF_SCTAB_HOST[207][0], PF_SCTAB_HOST[207][1]},
    {PF_SCTAB_HOST[208][0], PF_SCTAB_HOST[208][1]},
    {PF_SCTAB_HOST[209][0], PF_SCTAB_HOST[209][1]},
    {PF_SCTAB_HOST[210][0], PF_SCTAB_HOST[210][1]},
    {PF_SCTAB_HOST[211][0], PF_SCTAB_HOST[211][1]},
    {PF_SCTAB_HOST[212][0], PF_SCTAB_HOST[212][1]},
    {PF_SCTAB_HOST[213][0], PF_SCTAB_HOST[213][1]},
    {PF_SCTAB_HOST[214][0], PF_SCTAB_HOST[214][1]},
    {PF_SCTAB_HOST[215][0], PF_SCTAB_HOST[215][1]},
    {PF_SCTAB_HOST[216][0], PF_SCTAB_HOST[216][1]},
    {PF_SCTAB_HOST[217][0], PF_SCTAB_HOST[217][1]},
    {PF_SCTAB_HOST[218][0], PF_SCTAB_HOST[218][1]},
    {PF_SCTAB_HOST[219][0], PF_SCTAB_HOST[219][1]},
    {PF_SCTAB_HOST[220][0], PF_SCTAB_HOST[220][1]},
    {PF_SCTAB_HOST[221][0], PF_SCTAB_HOST[221][1]},
    {PF_SCTAB_HOST[222][0], PF_SCTAB_HOST[222][1]},
    {PF_SCTAB_HOST[223][0], PF_SCTAB_HOST[223][1]},
    {PF_SCTAB_HOST[224][0], PF_SCTAB_HOST[224][1]},
    {PF_SCTAB_HOST[225][0], PF_SCTAB_HOST[225][1]},
    {PF_SCTAB_HOST[226][0], PF_SCTAB_HOST[226][1]},
    {PF_SCTAB_HOST[227][0], PF_SCTAB_HOST[227][1]},
    {PF_SCTAB_HOST[228][0], PF_SCTAB_HOST[228][1]},
    {PF_SCTAB_HOST[229][0], PF_SCTAB_HOST[229][1]},
    {PF_SCTAB_HOST[230][0], PF_SCTAB_HOST[230][1]},
    {PF_SCTAB_HOST[231][0], PF_SCTAB_HOST[231][1]},
    {PF_SCTAB_HOST[232][0], PF_SCTAB_HOST[232][1]},
    {PF_SCTAB_HOST[233][0], PF_SCTAB_HOST[233][1]},
    {PF_SCTAB_HOST[234][0], PF_SCTAB_HOST[234][1]},
    {PF_SCTAB_HOST[235][0], PF_SCTAB_HOST[235][1]},
    {PF_SCTAB_HOST[236][0], PF_SCTAB_HOST[236][1]},
    {PF_SCTAB_HOST[237][0], PF_SCTAB_HOST[237][1]},
    {PF_SCTAB_HOST[238][0], PF_SCTAB_HOST[238][1]},
    {PF_SCTAB_HOST[239][0], PF_SCTAB_HOST[239][1]},
    {PF_SCTAB_HOST[240][0], PF_SCTAB_HOST[240][1]},
    {PF_SCTAB_HOST[241][0], PF_SCTAB_HOST[241][1]},
    {PF_SCTAB_HOST[242][0], PF_SCTAB_HOST[242][1]},
    {PF_SCTAB_HOST[243][0], PF_SCTAB_HOST[243][1]},
    {PF_SCTAB_HOST[244][0], PF_SCTAB_HOST[244][1]},
    {PF_SCTAB_HOST[245][0], PF_SCTAB_HOST[245][1]},
    {PF_SCTAB_HOST[246][0], PF_SCTAB_HOST[246][1]},
    {PF_SCTAB_HOST[247][0], PF_SCTAB_HOST[247][1]},
    {PF_SCTAB_HOST[248][0], PF_SCTAB_HOST[248][1]},
    {PF_SCTAB_HOST[249][0], PF_SCTAB_HOST[249][1]},
    {PF_SCTAB_HOST[250][0], PF_SCTAB_HOST[250][1]},
    {PF_SCTAB_HOST[251][0], PF_SCTAB_HOST[251][1]},
    {PF_SCTAB_HOST[252][0], PF_SCTAB_HOST[252][1]},
    {PF_SCTAB_HOST[253][0], PF_SCTAB_HOST[253][1]},
    {PF_SCTAB_HOST[254][0], PF_SCTAB_HOST[254][1]},
    {PF_SCTAB_HOST[255][0], PF_SCTAB_HOST[255][1]}
};
// copy the angle table into LDS (all threads of the block, then __syncthreads())
__device__ __forceinline__ void pf_sctab_load(double2 *tab) {
    for (int i = threadIdx.x; i < 256; i += blockDim.x) tab[i] = make_double2(PF_SCTAB_DEV[i][0], PF_SCTAB_DEV[i][1]);
}

// copy the table into LDS (call by all threads of the block, then __syncthreads())
__device__ __forceinline__ void pf_logtab_load(double2 *tab) {
    for (int i = threadIdx.x; i < 128; i += blockDim.x) tab[i] = make_double2(PF_LOGTAB_DEV[i][0], PF_LOGTAB_DEV[i][1]);
}

// -2 ln(u), u in (0,1) normal double
__device__ __forceinline__ double pf_neg2log_fast(double u, const double2 *tab) {
    const long long bits = __double_as_longlong(u);
    const int e = (int)(bits >> 52) - 1023;
    const int idx = (int)(bits >> 45) & 127;
    const double m = __longlong_as_double((bits & 0x000FFFFFFFFFFFFFll) | 0x3FF0000000000000ll);
    const double2 t = tab[idx];
    const double r = fma(m, t.x, -1.0);
    double p = fma(r, -1.0 / 7.0, 1.0 / 6.0);
    p = fma(r, p, -1.0 / 5.0);
    p = fma(r, p, 1.0 / 4.0);
    p = fma(r, p, -1.0 / 3.0);
    p = fma(r, p, 0.5);
    p = fma(r, p, -1.0);
    // -2 ln u = -2 (e ln2 + t.y) + 2 r p',  p' = -(1 - r/2 + r^2/3 - ...)  => ln(1+r) = -r p
    const double l = fma((double)e, 0.6931471805599453094, t.y);
    return fma(2.0 * r, p, -2.0 * l);
}

// sqrt(x), x in [1e-10, 100]
__device__ __forceinline__ double pf_sqrt_fast(double x) {
    double y = __builtin_amdgcn_rsq(x);          // ~2^-26 relative
    double g = x * y, h = 0.5 * y;
    double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    double d = fma(-g, g, x);
    return fma(d, h, g);
}

// sin(2 pi u), cos(2 pi u), u in (0,1)
__device__ __forceinline__ void pf_sincos2pi_fast(double u, double &s, double &c) {
    const double t4 = 4.0 * u;                    // quarter turns, exact
    const double q = rint(t4);
    const double f = 0.5 * (t4 - q);              // half-turn fraction in [-0.25, 0.25], exact
    const int iq = (int)q;
    const double f2 = f * f;
    double ps = -2.1915353447830217e-05;
    ps = fma(ps, f2, 0.00046630280576761255);
    ps = fma(ps, f2, -0.0073704309457143504);
    ps = fma(ps, f2, 0.08214588661112823);
    ps = fma(ps, f2, -0.5992645293207921);
    ps = fma(ps, f2, 2.5501640398773455);
    ps = fma(ps, f2, -5.16771278004997);
    ps = fma(ps, f2, 3.141592653589793);
    ps *= f;
    double pc = 4.303069587032947e-06;
    pc = fma(pc, f2, -0.0001046381049248457);
    pc = fma(pc, f2, 0.0019295743094039231);
    pc = fma(pc, f2, -0.02580689139001406);
    pc = fma(pc, f2, 0.2353306303588932);
    pc = fma(pc, f2, -1.3352627688545895);
    pc = fma(pc, f2, 4.0587121264167685);
    pc = fma(pc, f2, -4.934802200544679);
    pc = fma(pc, f2, 1.0);
    const bool swap = (iq & 1) != 0;
    double ss = swap ? pc : ps;
    double cc = swap ? ps : pc;
    // quadrant signs: iq = 0: (s, c); 1: (c, -s); 2: (-s, -c); 3: (-c, s); 4 == 0
    const long long sflip = ((long long)(iq & 2)) << 62;
    const long long cflip = ((long long)((iq + 1) & 2)) << 62;
    s = __longlong_as_double(__double_as_longlong(ss) ^ sflip);
    c = __longlong_as_double(__double_as_longlong(cc) ^ cflip);
}

__device__ __forceinline__ void pf_boxmuller4_fast(const uint32_t (&x)[4], const double2 *tab, double (&z)[4]) {
    const double S = 2.3283064365386962890625e-10;  // 2^-32
    const double u0 = ((double)x[0] + 0.5) * S, u1 = ((double)x[1] + 0.5) * S;
    const double u2 = ((double)x[2] + 0.5) * S, u3 = ((double)x[3] + 0.5) * S;
    const double r0 = pf_sqrt_fast(pf_neg2log_fast(u0, tab)), r1 = pf_sqrt_fast(pf_neg2log_fast(u2, tab));
    double s, c;
    pf_sincos2pi_fast(u1, s, c); z[0] = r0 * c; z[1] = r0 * s;
    pf_sincos2pi_fast(u3, s, c); z[2] = r1 * c; z[3] = r1 * s;
}

// same stream as pf_randn4 (pfmi_common.h), fast transcendental path
__device__ __forceinline__ void pf_randn4_fast(uint64_t seed, uint32_t g, uint32_t n, uint32_t stream,
                                               const double2 *tab, double (&z)[4]) {
    uint32_t x[4];
    pf_philox4x32_10(n, g, stream, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), x);
    pf_boxmuller4_fast(x, tab, z);
}

// ---- Box-Muller pair in seven pieces for kernels that interleave several independent chains (two pairs + the next
//      Philox call) inside one scheduling region.  Radius: -2 ln u via the log table; angle: theta = theta_i + delta with
//      {cos, sin}(theta_i) from a 256-entry LDS table and degree-5/6 Taylor polynomials in |delta| <= pi/256
//      (17 instead of 37 fp64 instructions for sin/cos).  Accurate to ~2e-16 absolute.
struct PfPair {
    double m, r, ty, p, v, g, h, rad, dlt, d2, cc, sc, sd, cs, sn;
    int e, idx, ai;
    __device__ __forceinline__ void s0(uint32_t xr, uint32_t xa) {
        const double S = 2.3283064365386962890625e-10;  // 2^-32
        const double ur = ((double)xr + 0.5) * S;
        const long long bits = __double_as_longlong(ur);
        e = (int)(bits >> 52) - 1023;
        idx = (int)(bits >> 45) & 127;
        m = __longlong_as_double((bits & 0x000FFFFFFFFFFFFFll) | 0x3FF0000000000000ll);
        ai = (int)(xa >> 24);
        // 2 pi u = theta_ai + delta,  delta = 2 pi ((xa mod 2^24) + 0.5 - 2^23) / 2^32
        dlt = ((double)((int)(xa & 0x00FFFFFFu) - 0x00800000) + 0.5) * 1.4629180792671596e-09;
    }
    __device__ __forceinline__ void s1(const double2 *ltab, const double2 *sctab) {
        const double2 t = ltab[idx];
        const double2 a = sctab[ai];
        ty = t.y;
        r = fma(m, t.x, -1.0);
        cc = a.x; sc = a.y;
        d2 = dlt * dlt;
    }
    __device__ __forceinline__ void s2() {
        double pp = fma(r, -1.0 / 7.0, 1.0 / 6.0);
        pp = fma(r, pp, -1.0 / 5.0);
        pp = fma(r, pp, 1.0 / 4.0);
        pp = fma(r, pp, -1.0 / 3.0);
        pp = fma(r, pp, 0.5);
        p = fma(r, pp, -1.0);
    }
    __device__ __forceinline__ void s3() {
        const double l = fma((double)e, 0.6931471805599453094, ty);
        v = fma(2.0 * r, p, -2.0 * l);
        const double y = __builtin_amdgcn_rsq(v);
        g = v * y;
        h = 0.5 * y;
    }
    __device__ __forceinline__ void s4() {
        const double rr = fma(-h, g, 0.5);
        g = fma(g, rr, g);
        h = fma(h, rr, h);
        const double dd = fma(-g, g, v);
        rad = fma(dd, h, g);
        double q = fma(d2, 1.0 / 120.0, -1.0 / 6.0);      // sin(delta) = delta (1 - d2/6 + d2^2/120)
        q = fma(q, d2, 1.0);
        sd = q * dlt;
    }
    __device__ __forceinline__ void s5() {
        double q = fma(d2, -1.0 / 720.0, 1.0 / 24.0);      // cos(delta) = 1 - d2/2 + d2^2/24 - d2^3/720
        q = fma(q, d2, -0.5);
        const double cd = fma(q, d2, 1.0);
        cs = fma(cc, cd, -(sc * sd));
        sn = fma(sc, cd, cc * sd);
    }
    __device__ __forceinline__ void s6(double &z0, double &z1) const {
        z0 = rad * cs;
        z1 = rad * sn;
    }
};
__device__ __forceinline__ void pf_philox_round(uint32_t &c0, uint32_t &c1, uint32_t &c2, uint32_t &c3,
                                                uint32_t &k0, uint32_t &k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
}
#endif

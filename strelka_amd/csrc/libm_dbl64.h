// libm_dbl64.h -- glibc's double-precision exp / log / log10 / log1p, restated for the device (see libm_flt32.h for the why).
//
// The SNV posteriors (calculate_result_set, calculate_result_set_grid, error_prob_to_qphred) are double-precision
// exp / log / log10 calls into the host C library.  glibc >= 2.28 (this image: 2.35): sysdeps/ieee754/dbl-64/e_exp.c and
// e_log.c are the table-driven routines of ARM's optimized-routines (128-entry tables, EXP_POLY_ORDER 5, LOG_POLY_ORDER 6,
// LOG_POLY1_ORDER 12), run in their FMA build on x86-64 CPUs with FMA; e_log10.c is the fdlibm wrapper around log, without
// fused operations.  Restated operation for operation below; oracle/libm_check.cpp compares them with the host libm.
// Tables: the published constants (__exp_data, __log_data), listed with tools/libm_tables.py.
//
// Each function returns false outside the restated domain (the caller then uses the device library): exp for nan and
// x >= 512; log / log10 for anything but positive normal numbers.
//
// PROVENANCE AND LICENCE.  Nothing here comes from /root/reference.  exp / log: ARM Optimized Routines (math/exp.c, log.c,
// exp_data.c, log_data.c; (c) Arm Limited; SPDX: MIT OR Apache-2.0 WITH LLVM-exception) as imported into the GNU C Library
// (sysdeps/ieee754/dbl-64/e_exp.c, e_log.c, e_exp_data.c, e_log_data.c; LGPL-2.1-or-later).  log10 / log1p: glibc's e_log10.c /
// s_log1p.c, from fdlibm ("Copyright (C) 1993 by Sun Microsystems, Inc. ... Permission to use, copy, modify, and distribute this
// software is freely granted, provided that this notice is preserved").  A restatement written for this repository (new code, the
// same operation sequence and published constants); redistribution should keep this notice and the licences named above.
#pragma once

#include "libm_flt32.h"

namespace sk_libm
{

/// { tail bits, scale bits } of 2^(i/128), i = 0..127 (__exp_data.tab).  Kernels that evaluate many exps per thread copy it
/// to LDS and pass that copy to exp_glibc.
SK_HD const uint64_t* exp_table()
{
    static constexpr uint64_t T[256] = {
        0x0000000000000000ull, 0x3ff0000000000000ull, 0x3c9b3b4f1a88bf6eull, 0x3feff63da9fb3335ull,
        0xbc7160139cd8dc5dull, 0x3fefec9a3e778061ull, 0xbc905e7a108766d1ull, 0x3fefe315e86e7f85ull,
        0x3c8cd2523567f613ull, 0x3fefd9b0d3158574ull, 0xbc8bce8023f98efaull, 0x3fefd06b29ddf6deull,
        0x3c60f74e61e6c861ull, 0x3fefc74518759bc8ull, 0x3c90a3e45b33d399ull, 0x3fefbe3ecac6f383ull,
        0x3c979aa65d837b6dull, 0x3fefb5586cf9890full, 0x3c8eb51a92fdeffcull, 0x3fefac922b7247f7ull,
        0x3c3ebe3d702f9cd1ull, 0x3fefa3ec32d3d1a2ull, 0xbc6a033489906e0bull, 0x3fef9b66affed31bull,
        0xbc9556522a2fbd0eull, 0x3fef9301d0125b51ull, 0xbc5080ef8c4eea55ull, 0x3fef8abdc06c31ccull,
        0xbc91c923b9d5f416ull, 0x3fef829aaea92de0ull, 0x3c80d3e3e95c55afull, 0x3fef7a98c8a58e51ull,
        0xbc801b15eaa59348ull, 0x3fef72b83c7d517bull, 0xbc8f1ff055de323dull, 0x3fef6af9388c8deaull,
        0x3c8b898c3f1353bfull, 0x3fef635beb6fcb75ull, 0xbc96d99c7611eb26ull, 0x3fef5be084045cd4ull,
        0x3c9aecf73e3a2f60ull, 0x3fef54873168b9aaull, 0xbc8fe782cb86389dull, 0x3fef4d5022fcd91dull,
        0x3c8a6f4144a6c38dull, 0x3fef463b88628cd6ull, 0x3c807a05b0e4047dull, 0x3fef3f49917ddc96ull,
        0x3c968efde3a8a894ull, 0x3fef387a6e756238ull, 0x3c875e18f274487dull, 0x3fef31ce4fb2a63full,
        0x3c80472b981fe7f2ull, 0x3fef2b4565e27cddull, 0xbc96b87b3f71085eull, 0x3fef24dfe1f56381ull,
        0x3c82f7e16d09ab31ull, 0x3fef1e9df51fdee1ull, 0xbc3d219b1a6fbffaull, 0x3fef187fd0dad990ull,
        0x3c8b3782720c0ab4ull, 0x3fef1285a6e4030bull, 0x3c6e149289cecb8full, 0x3fef0cafa93e2f56ull,
        0x3c834d754db0abb6ull, 0x3fef06fe0a31b715ull, 0x3c864201e2ac744cull, 0x3fef0170fc4cd831ull,
        0x3c8fdd395dd3f84aull, 0x3feefc08b26416ffull, 0xbc86a3803b8e5b04ull, 0x3feef6c55f929ff1ull,
        0xbc924aedcc4b5068ull, 0x3feef1a7373aa9cbull, 0xbc9907f81b512d8eull, 0x3feeecae6d05d866ull,
        0xbc71d1e83e9436d2ull, 0x3feee7db34e59ff7ull, 0xbc991919b3ce1b15ull, 0x3feee32dc313a8e5ull,
        0x3c859f48a72a4c6dull, 0x3feedea64c123422ull, 0xbc9312607a28698aull, 0x3feeda4504ac801cull,
        0xbc58a78f4817895bull, 0x3feed60a21f72e2aull, 0xbc7c2c9b67499a1bull, 0x3feed1f5d950a897ull,
        0x3c4363ed60c2ac11ull, 0x3feece086061892dull, 0x3c9666093b0664efull, 0x3feeca41ed1d0057ull,
        0x3c6ecce1daa10379ull, 0x3feec6a2b5c13cd0ull, 0x3c93ff8e3f0f1230ull, 0x3feec32af0d7d3deull,
        0x3c7690cebb7aafb0ull, 0x3feebfdad5362a27ull, 0x3c931dbdeb54e077ull, 0x3feebcb299fddd0dull,
        0xbc8f94340071a38eull, 0x3feeb9b2769d2ca7ull, 0xbc87deccdc93a349ull, 0x3feeb6daa2cf6642ull,
        0xbc78dec6bd0f385full, 0x3feeb42b569d4f82ull, 0xbc861246ec7b5cf6ull, 0x3feeb1a4ca5d920full,
        0x3c93350518fdd78eull, 0x3feeaf4736b527daull, 0x3c7b98b72f8a9b05ull, 0x3feead12d497c7fdull,
        0x3c9063e1e21c5409ull, 0x3feeab07dd485429ull, 0x3c34c7855019c6eaull, 0x3feea9268a5946b7ull,
        0x3c9432e62b64c035ull, 0x3feea76f15ad2148ull, 0xbc8ce44a6199769full, 0x3feea5e1b976dc09ull,
        0xbc8c33c53bef4da8ull, 0x3feea47eb03a5585ull, 0xbc845378892be9aeull, 0x3feea34634ccc320ull,
        0xbc93cedd78565858ull, 0x3feea23882552225ull, 0x3c5710aa807e1964ull, 0x3feea155d44ca973ull,
        0xbc93b3efbf5e2228ull, 0x3feea09e667f3bcdull, 0xbc6a12ad8734b982ull, 0x3feea012750bdabfull,
        0xbc6367efb86da9eeull, 0x3fee9fb23c651a2full, 0xbc80dc3d54e08851ull, 0x3fee9f7df9519484ull,
        0xbc781f647e5a3ecfull, 0x3fee9f75e8ec5f74ull, 0xbc86ee4ac08b7db0ull, 0x3fee9f9a48a58174ull,
        0xbc8619321e55e68aull, 0x3fee9feb564267c9ull, 0x3c909ccb5e09d4d3ull, 0x3feea0694fde5d3full,
        0xbc7b32dcb94da51dull, 0x3feea11473eb0187ull, 0x3c94ecfd5467c06bull, 0x3feea1ed0130c132ull,
        0x3c65ebe1abd66c55ull, 0x3feea2f336cf4e62ull, 0xbc88a1c52fb3cf42ull, 0x3feea427543e1a12ull,
        0xbc9369b6f13b3734ull, 0x3feea589994cce13ull, 0xbc805e843a19ff1eull, 0x3feea71a4623c7adull,
        0xbc94d450d872576eull, 0x3feea8d99b4492edull, 0x3c90ad675b0e8a00ull, 0x3feeaac7d98a6699ull,
        0x3c8db72fc1f0eab4ull, 0x3feeace5422aa0dbull, 0xbc65b6609cc5e7ffull, 0x3feeaf3216b5448cull,
        0x3c7bf68359f35f44ull, 0x3feeb1ae99157736ull, 0xbc93091fa71e3d83ull, 0x3feeb45b0b91ffc6ull,
        0xbc5da9b88b6c1e29ull, 0x3feeb737b0cdc5e5ull, 0xbc6c23f97c90b959ull, 0x3feeba44cbc8520full,
        0xbc92434322f4f9aaull, 0x3feebd829fde4e50ull, 0xbc85ca6cd7668e4bull, 0x3feec0f170ca07baull,
        0x3c71affc2b91ce27ull, 0x3feec49182a3f090ull, 0x3c6dd235e10a73bbull, 0x3feec86319e32323ull,
        0xbc87c50422622263ull, 0x3feecc667b5de565ull, 0x3c8b1c86e3e231d5ull, 0x3feed09bec4a2d33ull,
        0xbc91bbd1d3bcbb15ull, 0x3feed503b23e255dull, 0x3c90cc319cee31d2ull, 0x3feed99e1330b358ull,
        0x3c8469846e735ab3ull, 0x3feede6b5579fdbfull, 0xbc82dfcd978e9db4ull, 0x3feee36bbfd3f37aull,
        0x3c8c1a7792cb3387ull, 0x3feee89f995ad3adull, 0xbc907b8f4ad1d9faull, 0x3feeee07298db666ull,
        0xbc55c3d956dcaebaull, 0x3feef3a2b84f15fbull, 0xbc90a40e3da6f640ull, 0x3feef9728de5593aull,
        0xbc68d6f438ad9334ull, 0x3feeff76f2fb5e47ull, 0xbc91eee26b588a35ull, 0x3fef05b030a1064aull,
        0x3c74ffd70a5fddcdull, 0x3fef0c1e904bc1d2ull, 0xbc91bdfbfa9298acull, 0x3fef12c25bd71e09ull,
        0x3c736eae30af0cb3ull, 0x3fef199bdd85529cull, 0x3c8ee3325c9ffd94ull, 0x3fef20ab5fffd07aull,
        0x3c84e08fd10959acull, 0x3fef27f12e57d14bull, 0x3c63cdaf384e1a67ull, 0x3fef2f6d9406e7b5ull,
        0x3c676b2c6c921968ull, 0x3fef3720dcef9069ull, 0xbc808a1883ccb5d2ull, 0x3fef3f0b555dc3faull,
        0xbc8fad5d3ffffa6full, 0x3fef472d4a07897cull, 0xbc900dae3875a949ull, 0x3fef4f87080d89f2ull,
        0x3c74a385a63d07a7ull, 0x3fef5818dcfba487ull, 0xbc82919e2040220full, 0x3fef60e316c98398ull,
        0x3c8e5a50d5c192acull, 0x3fef69e603db3285ull, 0x3c843a59ac016b4bull, 0x3fef7321f301b460ull,
        0xbc82d52107b43e1full, 0x3fef7c97337b9b5full, 0xbc892ab93b470dc9ull, 0x3fef864614f5a129ull,
        0x3c74b604603a88d3ull, 0x3fef902ee78b3ff6ull, 0x3c83c5ec519d7271ull, 0x3fef9a51fbc74c83ull,
        0xbc8ff7128fd391f0ull, 0x3fefa4afa2a490daull, 0xbc8dae98e223747dull, 0x3fefaf482d8e67f1ull,
        0x3c8ec3bc41aa2008ull, 0x3fefba1bee615a27ull, 0x3c842b94c3a9eb32ull, 0x3fefc52b376bba97ull,
        0x3c8a64a931d185eeull, 0x3fefd0765b6e4540ull, 0xbc8e37bae43be3edull, 0x3fefdbfdad9cbe14ull,
        0x3c77893b4d91cd9dull, 0x3fefe7c1819e90d8ull, 0x3c5305c14160cc89ull, 0x3feff3c22b8f71f1ull };
    return T;
}

/// { 1/c, log(c) } for the 128 sub-intervals (__log_data.tab), flattened
SK_HD const double* log_table()
{
    static constexpr double T[256] = {
        0x1.734f0c3e0de9fp+0, -0x1.7cc7f79e69000p-2, 0x1.713786a2ce91fp+0, -0x1.76feec20d0000p-2,
        0x1.6f26008fab5a0p+0, -0x1.713e31351e000p-2, 0x1.6d1a61f138c7dp+0, -0x1.6b85b38287800p-2,
        0x1.6b1490bc5b4d1p+0, -0x1.65d5590807800p-2, 0x1.69147332f0cbap+0, -0x1.602d076180000p-2,
        0x1.6719f18224223p+0, -0x1.5a8ca86909000p-2, 0x1.6524f99a51ed9p+0, -0x1.54f4356035000p-2,
        0x1.63356aa8f24c4p+0, -0x1.4f637c36b4000p-2, 0x1.614b36b9ddc14p+0, -0x1.49da7fda85000p-2,
        0x1.5f66452c65c4cp+0, -0x1.445923989a800p-2, 0x1.5d867b5912c4fp+0, -0x1.3edf439b0b800p-2,
        0x1.5babccb5b90dep+0, -0x1.396ce448f7000p-2, 0x1.59d61f2d91a78p+0, -0x1.3401e17bda000p-2,
        0x1.5805612465687p+0, -0x1.2e9e2ef468000p-2, 0x1.56397cee76bd3p+0, -0x1.2941b3830e000p-2,
        0x1.54725e2a77f93p+0, -0x1.23ec58cda8800p-2, 0x1.52aff42064583p+0, -0x1.1e9e129279000p-2,
        0x1.50f22dbb2bddfp+0, -0x1.1956d2b48f800p-2, 0x1.4f38f4734ded7p+0, -0x1.141679ab9f800p-2,
        0x1.4d843cfde2840p+0, -0x1.0edd094ef9800p-2, 0x1.4bd3ec078a3c8p+0, -0x1.09aa518db1000p-2,
        0x1.4a27fc3e0258ap+0, -0x1.047e65263b800p-2, 0x1.4880524d48434p+0, -0x1.feb224586f000p-3,
        0x1.46dce1b192d0bp+0, -0x1.f474a7517b000p-3, 0x1.453d9d3391854p+0, -0x1.ea4443d103000p-3,
        0x1.43a2744b4845ap+0, -0x1.e020d44e9b000p-3, 0x1.420b54115f8fbp+0, -0x1.d60a22977f000p-3,
        0x1.40782da3ef4b1p+0, -0x1.cc00104959000p-3, 0x1.3ee8f5d57fe8fp+0, -0x1.c202956891000p-3,
        0x1.3d5d9a00b4ce9p+0, -0x1.b81178d811000p-3, 0x1.3bd60c010c12bp+0, -0x1.ae2c9ccd3d000p-3,
        0x1.3a5242b75dab8p+0, -0x1.a45402e129000p-3, 0x1.38d22cd9fd002p+0, -0x1.9a877681df000p-3,
        0x1.3755bc5847a1cp+0, -0x1.90c6d69483000p-3, 0x1.35dce49ad36e2p+0, -0x1.87120a645c000p-3,
        0x1.34679984dd440p+0, -0x1.7d68fb4143000p-3, 0x1.32f5cceffcb24p+0, -0x1.73cb83c627000p-3,
        0x1.3187775a10d49p+0, -0x1.6a39a9b376000p-3, 0x1.301c8373e3990p+0, -0x1.60b3154b7a000p-3,
        0x1.2eb4ebb95f841p+0, -0x1.5737d76243000p-3, 0x1.2d50a0219a9d1p+0, -0x1.4dc7b8fc23000p-3,
        0x1.2bef9a8b7fd2ap+0, -0x1.4462c51d20000p-3, 0x1.2a91c7a0c1babp+0, -0x1.3b08abc830000p-3,
        0x1.293726014b530p+0, -0x1.31b996b490000p-3, 0x1.27dfa5757a1f5p+0, -0x1.2875490a44000p-3,
        0x1.268b39b1d3bbfp+0, -0x1.1f3b9f879a000p-3, 0x1.2539d838ff5bdp+0, -0x1.160c8252ca000p-3,
        0x1.23eb7aac9083bp+0, -0x1.0ce7f57f72000p-3, 0x1.22a012ba940b6p+0, -0x1.03cdc49fea000p-3,
        0x1.2157996cc4132p+0, -0x1.f57bdbc4b8000p-4, 0x1.201201dd2fc9bp+0, -0x1.e370896404000p-4,
        0x1.1ecf4494d480bp+0, -0x1.d17983ef94000p-4, 0x1.1d8f5528f6569p+0, -0x1.bf9674ed8a000p-4,
        0x1.1c52311577e7cp+0, -0x1.adc79202f6000p-4, 0x1.1b17c74cb26e9p+0, -0x1.9c0c3e7288000p-4,
        0x1.19e010c2c1ab6p+0, -0x1.8a646b372c000p-4, 0x1.18ab07bb670bdp+0, -0x1.78d01b3ac0000p-4,
        0x1.1778a25efbcb6p+0, -0x1.674f145380000p-4, 0x1.1648d354c31dap+0, -0x1.55e0e6d878000p-4,
        0x1.151b990275fddp+0, -0x1.4485cdea1e000p-4, 0x1.13f0ea432d24cp+0, -0x1.333d94d6aa000p-4,
        0x1.12c8b7210f9dap+0, -0x1.22079f8c56000p-4, 0x1.11a3028ecb531p+0, -0x1.10e4698622000p-4,
        0x1.107fbda8434afp+0, -0x1.ffa6c6ad20000p-5, 0x1.0f5ee0f4e6bb3p+0, -0x1.dda8d4a774000p-5,
        0x1.0e4065d2a9fcep+0, -0x1.bbcece4850000p-5, 0x1.0d244632ca521p+0, -0x1.9a1894012c000p-5,
        0x1.0c0a77ce2981ap+0, -0x1.788583302c000p-5, 0x1.0af2f83c636d1p+0, -0x1.5715e67d68000p-5,
        0x1.09ddb98a01339p+0, -0x1.35c8a49658000p-5, 0x1.08cabaf52e7dfp+0, -0x1.149e364154000p-5,
        0x1.07b9f2f4e28fbp+0, -0x1.e72c082eb8000p-6, 0x1.06ab58c358f19p+0, -0x1.a55f152528000p-6,
        0x1.059eea5ecf92cp+0, -0x1.63d62cf818000p-6, 0x1.04949cdd12c90p+0, -0x1.228fb8caa0000p-6,
        0x1.038c6c6f0ada9p+0, -0x1.c317b20f90000p-7, 0x1.02865137932a9p+0, -0x1.419355daa0000p-7,
        0x1.0182427ea7348p+0, -0x1.81203c2ec0000p-8, 0x1.008040614b195p+0, -0x1.0040979240000p-9,
        0x1.fe01ff726fa1ap-1, 0x1.feff384900000p-9, 0x1.fa11cc261ea74p-1, 0x1.7dc41353d0000p-7,
        0x1.f6310b081992ep-1, 0x1.3cea3c4c28000p-6, 0x1.f25f63ceeadcdp-1, 0x1.b9fc114890000p-6,
        0x1.ee9c8039113e7p-1, 0x1.1b0d8ce110000p-5, 0x1.eae8078cbb1abp-1, 0x1.58a5bd001c000p-5,
        0x1.e741aa29d0c9bp-1, 0x1.95c8340d88000p-5, 0x1.e3a91830a99b5p-1, 0x1.d276aef578000p-5,
        0x1.e01e009609a56p-1, 0x1.07598e598c000p-4, 0x1.dca01e577bb98p-1, 0x1.253f5e30d2000p-4,
        0x1.d92f20b7c9103p-1, 0x1.42edd8b380000p-4, 0x1.d5cac66fb5ccep-1, 0x1.606598757c000p-4,
        0x1.d272caa5ede9dp-1, 0x1.7da76356a0000p-4, 0x1.cf26e3e6b2ccdp-1, 0x1.9ab434e1c6000p-4,
        0x1.cbe6da2a77902p-1, 0x1.b78c7bb0d6000p-4, 0x1.c8b266d37086dp-1, 0x1.d431332e72000p-4,
        0x1.c5894bd5d5804p-1, 0x1.f0a3171de6000p-4, 0x1.c26b533bb9f8cp-1, 0x1.067152b914000p-3,
        0x1.bf583eeece73fp-1, 0x1.147858292b000p-3, 0x1.bc4fd75db96c1p-1, 0x1.2266ecdca3000p-3,
        0x1.b951e0c864a28p-1, 0x1.303d7a6c55000p-3, 0x1.b65e2c5ef3e2cp-1, 0x1.3dfc33c331000p-3,
        0x1.b374867c9888bp-1, 0x1.4ba366b7a8000p-3, 0x1.b094b211d304ap-1, 0x1.5933928d1f000p-3,
        0x1.adbe885f2ef7ep-1, 0x1.66acd2418f000p-3, 0x1.aaf1d31603da2p-1, 0x1.740f8ec669000p-3,
        0x1.a82e63fd358a7p-1, 0x1.815c0f51af000p-3, 0x1.a5740ef09738bp-1, 0x1.8e92954f68000p-3,
        0x1.a2c2a90ab4b27p-1, 0x1.9bb3602f84000p-3, 0x1.a01a01393f2d1p-1, 0x1.a8bed1c2c0000p-3,
        0x1.9d79f24db3c1bp-1, 0x1.b5b515c01d000p-3, 0x1.9ae2505c7b190p-1, 0x1.c2967ccbcc000p-3,
        0x1.9852ef297ce2fp-1, 0x1.cf635d5486000p-3, 0x1.95cbaeea44b75p-1, 0x1.dc1bd3446c000p-3,
        0x1.934c69de74838p-1, 0x1.e8c01b8cfe000p-3, 0x1.90d4f2f6752e6p-1, 0x1.f5509c0179000p-3,
        0x1.8e6528effd79dp-1, 0x1.00e6c121fb800p-2, 0x1.8bfce9fcc007cp-1, 0x1.071b80e93d000p-2,
        0x1.899c0dabec30ep-1, 0x1.0d46b9e867000p-2, 0x1.87427aa2317fbp-1, 0x1.13687334bd000p-2,
        0x1.84f00acb39a08p-1, 0x1.1980d67234800p-2, 0x1.82a49e8653e55p-1, 0x1.1f8ffe0cc8000p-2,
        0x1.8060195f40260p-1, 0x1.2595fd7636800p-2, 0x1.7e22563e0a329p-1, 0x1.2b9300914a800p-2,
        0x1.7beb377dcb5adp-1, 0x1.3187210436000p-2, 0x1.79baa679725c2p-1, 0x1.377266dec1800p-2,
        0x1.77907f2170657p-1, 0x1.3d54ffbaf3000p-2, 0x1.756cadbd6130cp-1, 0x1.432eee32fe000p-2 };
    return T;
}

SK_HD bool exp_glibc(const double x, double& out, const uint64_t* T = exp_table())
{
    constexpr double InvLn2N = 0x1.71547652b82fep+7, Shift = 0x1.8000000000000p+52, NegLn2hiN = -0x1.62e42fefa0000p-8, NegLn2loN = -0x1.cf79abc9e3b3ap-47;
    constexpr double C2 = 0x1.ffffffffffdbdp-2, C3 = 0x1.555555555543cp-3, C4 = 0x1.55555cf172b91p-5, C5 = 0x1.1111167a4d017p-7;
    if (!(x > -512.0 && x < 512.0)) { // rare: the common range runs straight through
        if (!(x < 512.0)) return false; // overflow range, nan
        if (x <= -1024.0) {             // underflow (includes -inf)
            out = 0.0;
            return true;
        }
    }
    // (|x| < 2^-54 needs no special case here: the main path gives fma(1, x, 1) = 1.0 + x as well)
    // x = ln2/128 * k + r
    double kd = fma_(InvLn2N, x, Shift);
    const uint64_t ki = as_u64(kd);
    kd -= Shift;
    double r = fma_(kd, NegLn2hiN, x);
    r = fma_(kd, NegLn2loN, r);
    const unsigned idx = 2u * unsigned(ki % 128u);
    const uint64_t top = ki << (52 - 7);
    const double tail = as_f64(T[idx]);
    uint64_t sbits = T[idx + 1] + top;
    const double r2 = r * r;
    double tmp = tail + r;
    tmp = fma_(r2, fma_(r, C3, C2), tmp);
    tmp = fma_(r2 * r2, fma_(r, C5, C4), tmp);
    if (x > -512.0) {
        const double scale = as_f64(sbits);
        out = fma_(scale, tmp, scale);
        return true;
    }
    // specialcase(), k < 0: the result may be subnormal
    sbits += 1022ull << 52;
    const double scale = as_f64(sbits);
    double y = scale + scale * tmp; // (the out-of-line special case of the FMA build keeps these two unfused)
    if (y < 1.0) {
        double lo = scale - y + scale * tmp;
        const double hi = 1.0 + y;
        lo = 1.0 - hi + y + lo;
        y = (hi + lo) - 1.0;
        if (y == 0.0) y = 0.0;
    }
    out = 0x1p-1022 * y;
    return true;
}

SK_HD bool log_glibc(const double x, double& out, const double* T = log_table())
{
    constexpr double Ln2hi = 0x1.62e42fefa3800p-1, Ln2lo = 0x1.ef35793c76730p-45;
    constexpr double A[5] = { -0x1.0000000000001p-1, 0x1.555555551305bp-2, -0x1.fffffffeb4590p-3, 0x1.999b324f10111p-3, -0x1.55575e506c89fp-3 };
    constexpr double B[11] = { -0x1.0000000000000p-1, 0x1.5555555555577p-2, -0x1.ffffffffffdcbp-3, 0x1.999999995dd0cp-3, -0x1.55555556745a7p-3, 0x1.24924a344de30p-3, -0x1.fffffa4423d65p-4, 0x1.c7184282ad6cap-4, -0x1.999eb43b068ffp-4, 0x1.78182f7afd085p-4, -0x1.5521375d145cdp-4 };
    const uint64_t ix = as_u64(x);
    const uint64_t LO = as_u64(1.0 - 0x1p-4), HI = as_u64(1.0 + 0x1.09p-4);
    if (ix - LO < HI - LO) { // close to 1
        if (ix == as_u64(1.0)) {
            out = 0.0;
            return true;
        }
        const double r = x - 1.0, r2 = r * r, r3 = r * r2;
        const double p3 = fma_(r3, B[10], fma_(r2, B[9], fma_(r, B[8], B[7])));
        const double p2 = fma_(r3, p3, fma_(r2, B[6], fma_(r, B[5], B[4])));
        const double p1 = fma_(r3, p2, fma_(r2, B[3], fma_(r, B[2], B[1])));
        double w = r * 0x1p27;
        const double rhi = r + w - w;
        const double rlo = r - rhi;
        w = rhi * rhi * B[0]; // B[0] == -0.5
        const double hi = r + w;
        double lo = r - hi + w;
        lo = fma_(B[0] * rlo, rhi + r, lo);
        double y = fma_(r3, p1, lo);
        y += hi;
        out = y;
        return true;
    }
    const uint32_t top = uint32_t(ix >> 48);
    if (top - 0x0010u >= 0x7ff0u - 0x0010u) return false; // zero, subnormal, negative, inf, nan
    const uint64_t tmp = ix - 0x3fe6000000000000ull;
    const int i = int((tmp >> (52 - 7)) % 128u);
    const int64_t k = int64_t(tmp) >> 52; // arithmetic shift
    const uint64_t iz = ix - (tmp & (0xfffull << 52));
    const double invc = T[2 * i], logc = T[2 * i + 1];
    const double z = as_f64(iz);
    const double r = fma_(z, invc, -1.0);
    const double kd = double(k);
    const double w = fma_(kd, Ln2hi, logc);
    const double hi = w + r;
    const double lo = fma_(kd, Ln2lo, w - hi + r);
    const double r2 = r * r;
    const double q = fma_(r2, fma_(r, A[4], A[3]), fma_(r, A[2], A[1]));
    double y = fma_(r2, A[0], lo);
    y = fma_(r * r2, q, y);
    out = y + hi;
    return true;
}

/// e_log10.c: log10(x) = k log10(2) + log(mantissa) / ln(10), split constants, no fused operations
SK_HD bool log10_glibc(double x, double& out, const double* T = log_table())
{
    uint64_t u = as_u64(x);
    int32_t hx = int32_t(u >> 32);
    if (hx < 0x00100000 || hx >= 0x7ff00000) return false; // zero, subnormal, negative, inf, nan
    int32_t k = (hx >> 20) - 1023;
    const int32_t i = int32_t((uint32_t(k) & 0x80000000u) >> 31);
    hx = (hx & 0x000fffff) | ((0x3ff - i) << 20);
    const double y = double(k + i);
    u = (u & 0xffffffffull) | (uint64_t(uint32_t(hx)) << 32);
    x = as_f64(u);
    double lg;
    if (!log_glibc(x, lg, T)) return false;
    const double ivln10 = as_f64(0x3FDBCB7B1526E50Eull), log10_2hi = as_f64(0x3FD34413509F6000ull),
                 log10_2lo = as_f64(0x3D59FEF311F12B36ull);
    const double z = y * log10_2lo + ivln10 * lg;
    out = z + y * log10_2hi;
    return true;
}

/// glibc log1p (s_log1p.c: the fdlibm routine with glibc's split polynomial evaluation; no FMA build) for 0 <= x < 0.41422.
/// Plain double arithmetic: callers compile with -ffp-contract=off.
SK_HD bool log1p_glibc(const double x, double& out)
{
    const int32_t hx = int32_t(as_u64(x) >> 32);
    if (hx < 0 || hx >= 0x3FDA827A) return false; // negative, >= 0.41422, inf, nan
    if (hx < 0x3e200000) {                        // x < 2^-29
        out = (hx < 0x3c900000) ? x : x - x * x * 0.5;
        return true;
    }
    const double Lp1 = as_f64(0x3FE5555555555593ull), Lp2 = as_f64(0x3FD999999997FA04ull), Lp3 = as_f64(0x3FD2492494229359ull),
                 Lp4 = as_f64(0x3FCC71C51D8E78AFull), Lp5 = as_f64(0x3FC7466496CB03DEull), Lp6 = as_f64(0x3FC39A09D078C69Full),
                 Lp7 = as_f64(0x3FC2F112DF3E5244ull);
    const double f = x;
    const double hfsq = 0.5 * f * f;
    const double s = f / (2.0 + f);
    const double z = s * s;
    const double R1 = z * Lp1, z2 = z * z;
    const double R2 = Lp2 + z * Lp3, z4 = z2 * z2;
    const double R3 = Lp4 + z * Lp5, z6 = z4 * z2;
    const double R4 = Lp6 + z * Lp7;
    const double R = R1 + z2 * R2 + z4 * R3 + z6 * R4;
    out = f - (hfsq - s * (hfsq + R));
    return true;
}

} // namespace sk_libm

#if defined(__HIPCC__)
// what the kernels call: the restated routine when the host libm is the implementation restated above (sk_init checks),
// the device library's otherwise and outside the restated domain
struct SkLibmTables // where the kernels read the two tables from: constant memory by default, or a block's LDS copy
{
    const uint64_t* exp_t;
    const double* log_t;
};
__device__ __forceinline__ SkLibmTables sk_libm_tables_default() { return SkLibmTables{ sk_libm::exp_table(), sk_libm::log_table() }; }
/// copy both tables into `lds` (512 x 8 bytes) with the whole block; the caller synchronises
__device__ __forceinline__ SkLibmTables sk_libm_tables_to_lds(uint64_t* lds, const int tid, const int nthreads)
{
    const uint64_t* e = sk_libm::exp_table();
    const double* l = sk_libm::log_table();
    for (int i = tid; i < 256; i += nthreads) {
        lds[i] = e[i];
        lds[256 + i] = sk_libm::as_u64(l[i]);
    }
    return SkLibmTables{ lds, reinterpret_cast<const double*>(lds + 256) };
}
__device__ __forceinline__ double sk_exp(const double x, const int exact_libm, const SkLibmTables& t)
{
    double r;
    return (exact_libm && sk_libm::exp_glibc(x, r, t.exp_t)) ? r : exp(x);
}
__device__ __forceinline__ double sk_log(const double x, const int exact_libm, const SkLibmTables& t)
{
    double r;
    return (exact_libm && sk_libm::log_glibc(x, r, t.log_t)) ? r : log(x);
}
__device__ __forceinline__ double sk_log10(const double x, const int exact_libm, const SkLibmTables& t)
{
    double r;
    return (exact_libm && sk_libm::log10_glibc(x, r, t.log_t)) ? r : log10(x);
}
__device__ __forceinline__ double sk_log1p(const double x, const int exact_libm)
{
    double r;
    return (exact_libm && sk_libm::log1p_glibc(x, r)) ? r : log1p(x);
}
// Out-of-line twins for the somatic grid posterior: it unrolls ~150 call sites so that every likelihood index is static,
// and inlining a table-driven routine into each of them makes the compiler give the unrolling up and index the
// likelihood arrays in scratch memory (measured: 0.70 ms instead of 0.50 ms per 2^20 loci).
__device__ __noinline__ double sk_exp_call(const double x, const int exact_libm, const SkLibmTables& t) { return sk_exp(x, exact_libm, t); }
__device__ __noinline__ double sk_log_call(const double x, const int exact_libm, const SkLibmTables& t) { return sk_log(x, exact_libm, t); }
__device__ __noinline__ double sk_log10_call(const double x, const int exact_libm, const SkLibmTables& t) { return sk_log10(x, exact_libm, t); }
#endif

/* dn_action_sat.h -- the saturation constants of the float32 action chain (A1-A3), shared by the HIP kernels
 * (dn_kernels.hip: rotor_force_sat) and by the exhaustive checker tests/tools/check_action_chain_exact.c, which
 * PROVES them against the literal numpy-order chain for every float32 action (all 2^32 bit patterns but the NaNs).
 *
 * PBDroneEnv._preprocessAction clips the (rescaled) action to [a_low, a_high] (PBDroneEnv.py:889) before anything
 * else reads it, so the rotor force / torque are functions of that clipped thrust alone: an action whose rescaled
 * value is <= a_low gives the force and torque of a_low, one >= a_high those of a_high.  PBDroneEnv.rescale_action
 * (PBDroneEnv.py:949-971) is monotone in the action (a subtraction of, a correctly rounded division by and a
 * multiplication with positive constants, an addition), so "rescaled value <= a_low" is "raw action <= one float32
 * threshold", and likewise at the top.  99.6 % of U(-1,1) actions are outside the band in between.
 *
 *   raw action a (normalize_actions = 1):   a <= DN_ACT_SAT_LO  ->  (F_LO, TQ_LO);   a >= DN_ACT_SAT_HI  ->  (F_HI, TQ_HI)
 *   command c    (normalize_actions = 0):   c <= DN_A_LOW       ->  (F_LO, TQ_LO);   c >= DN_A_HIGH      ->  (F_HI, TQ_HI)
 *
 * IEEE-754 binary32 bit patterns (plain C: no hexadecimal float literals needed).
 */
#ifndef DN_ACTION_SAT_H
#define DN_ACTION_SAT_H

#define DN_A_LOW_BITS 0x3ce6b357u        /* a_low  = float32(KF (0.2685 * 20000 + 4070.3)^2) = 0.02816169,  PBDroneEnv.py:113-116 */
#define DN_A_HIGH_BITS 0x3e17e6d2u       /* a_high = float32(KF (0.2685 * 65535 + 4070.3)^2) = 0.14834145 */
#define DN_ACT_SAT_LO_BITS 0x3db83474u   /* 0.0899437964: the largest action whose rescaled value is <= a_low */
#define DN_ACT_SAT_HI_BITS 0x3dc6fea6u   /* 0.0971653908: the smallest action whose rescaled value is >= a_high */
#define DN_F_LO_BITS 0x3ce6b357u         /* rotor force at a_low  = (0.2685 * 20000 + 4070.3)^2 KF in float32 (it round-trips to a_low itself) */
#define DN_TQ_LO_BITS 0x3a397eb3u        /* rotor torque at a_low */
#define DN_F_HI_BITS 0x3e17e6d2u         /* rotor force at a_high */
#define DN_TQ_HI_BITS 0x3b7445f2u        /* rotor torque at a_high */

#endif

// hrfd_shim_layout.h -- why the shim classes are small, and the numbers that say how small they must be.
//
// The reference application is compiled as ONE g++ command over its .cc files (radioDiags/buildRadioDiags.sh:50)
// and hdr_diags/Radio.h:16,18,28 pull in "IqDataProcessor.h", "DataProvider.h" and "BasebandDataProcessor.h" with
// QUOTED includes.  A quoted include looks in the directory of the including file first; -I comes second.  So
// Radio.cc, diagUi.cc, FrequencyScanner.cc, ... see the REFERENCE's declarations of those three classes as long as the
// reference's headers lie beside Radio.h, whatever the -I order, and `new IqDataProcessor(...)` in Radio.cc:175
// allocates sizeof(reference class) bytes for an object whose member functions are the shim's.
//
// That is made harmless instead of forbidden: every shim class is LAYOUT-CONTAINED in the reference class of the
// same name -- no larger, no stricter alignment, no virtual functions, no inline member functions on either side (the
// reference's headers have none: every member function is an out-of-line symbol that the shim defines) -- so an object
// allocated by a translation unit that saw the reference's header is big enough for everything the shim's member
// functions touch.  Anything that does not fit lives in a heap block owned by the object.
//
// The numbers: sizeof / alignof of the reference classes on x86-64 Linux (g++ 11, the reference's headers as they lie
// in radioDiags/ at the surveyed commit).  tests/test_dropin.py recomputes them from the reference's headers whenever
// /root/reference is present and fails if one of them has moved.
#ifndef HRFD_SHIM_LAYOUT_H
#define HRFD_SHIM_LAYOUT_H

#define HRFD_REF_SIZEOF_IqDataProcessor        32920   /* hdr_diags/IqDataProcessor.h:17-127 */
#define HRFD_REF_SIZEOF_BasebandDataProcessor  17664   /* hdr_diags/BasebandDataProcessor.h */
#define HRFD_REF_SIZEOF_DataProvider             272   /* hdr_diags/DataProvider.h */
#define HRFD_REF_SIZEOF_AmDemodulator           4168   /* AmDemodulator/AmDemodulator.h */
#define HRFD_REF_SIZEOF_FmDemodulator          33848   /* FmDemodulator/FmDemodulator.h */
#define HRFD_REF_SIZEOF_WbFmDemodulator        66608   /* WbFmDemodulator/WbFmDemodulator.h */
#define HRFD_REF_SIZEOF_SsbDemodulator          4184   /* SsbDemodulator/SsbDemodulator.h */
#define HRFD_REF_SIZEOF_AmModulator             4224   /* AmModulator/AmModulator.h */
#define HRFD_REF_SIZEOF_FmModulator             4232   /* FmModulator/FmModulator.h */
#define HRFD_REF_SIZEOF_WbFmModulator         100320   /* WbFmModulator/WbFmModulator.h */
#define HRFD_REF_SIZEOF_SsbModulator            4240   /* SsbModulator/SsbModulator.h */
#define HRFD_REF_SIZEOF_Nco                   131088   /* Nco/Nco.h */
#define HRFD_REF_ALIGNOF_ALL                       8

// Used in hrfd_shim.cc, where every shim class is complete.  (When hrfd_shim.cc itself is compiled against a
// reference header by accident, sizeof equals the reference's and the assertion holds trivially -- but then the
// shim's member definitions do not compile, so that mistake cannot go unnoticed either.)
#define HRFD_SHIM_FITS(T)                                                                              \
  static_assert(sizeof(T) <= HRFD_REF_SIZEOF_##T,                                                      \
                #T ": the shim class must fit inside the reference class (hrfd_shim_layout.h)");       \
  static_assert(alignof(T) <= HRFD_REF_ALIGNOF_ALL, #T ": alignment above the reference class's");     \
  static_assert(!__is_polymorphic(T), #T ": the reference class has no vtable, neither may the shim's")

#endif

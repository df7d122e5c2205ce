// tests/devsim (TEST INFRASTRUCTURE): kdev.h's three system-scope packet accessors for a host compiler; included by kdev.h
// inside namespace xmpi { namespace { ... } }, after pack_t, when XMPI_DEVSIM is defined.
//
// A load is two 8-byte atomic loads (what the LL lines rely on: each half is either the old line or the new one).  A store
// is a PLAIN 16-byte store: the stepped kernels order their data by flag words only, so a reader the flags have not ordered
// behind the store is reported by ThreadSanitizer as the data race it is.
__device__ __forceinline__ void ld_sys128_issue(pack_t& v, const pack_t* p) {
  ::devsim::sync_point();
  const uint64_t* q = reinterpret_cast<const uint64_t*>(p);
  const uint64_t lo = __atomic_load_n(q, __ATOMIC_ACQUIRE), hi = __atomic_load_n(q + 1, __ATOMIC_ACQUIRE);
  v = pack_t{(unsigned)lo, (unsigned)(lo >> 32), (unsigned)hi, (unsigned)(hi >> 32)};
  if (::devsim::g_corrupt_armed) ::devsim::corrupt_bits(p, &v, 16, ::devsim::ACC_SYS_LOAD);  // (DEVSIM_CORRUPT_FORM: see hip_runtime.h)
}
__device__ __forceinline__ void st_sys128(pack_t* p, pack_t v) {
  if (::devsim::g_corrupt_armed) ::devsim::corrupt_bits(p, &v, 16, ::devsim::ACC_SYS_STORE);
  *p = v;
}
template <int U>
__device__ __forceinline__ void sys128_wait(pack_t (&)[U]) {}

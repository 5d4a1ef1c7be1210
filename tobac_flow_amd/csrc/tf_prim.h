// Device-wide primitives (scan, run-length encode, radix sort, reduce by key): rocPRIM called directly (round 5; rounds 1 - 4
// went through hipcub, the CUB-shaped wrapper around the same library).  Thin inline forms with the argument order the
// call sites use: (temp storage, its size in / out, ..., item count, stream).  A null temp pointer = size query.
#pragma once
#include <cstring>
#include <iterator>
#include <rocprim/rocprim.hpp>

template <typename In, typename Out>
static inline hipError_t tf_exclusive_sum(void *tmp, size_t &bytes, In in, Out out, size_t n, hipStream_t s = 0)
{
    typedef typename std::iterator_traits<Out>::value_type T;
    return rocprim::exclusive_scan(tmp, bytes, in, out, T(0), n, rocprim::plus<T>(), s);
}
template <typename In, typename Out, typename T>
static inline hipError_t tf_exclusive_sum_from(void *tmp, size_t &bytes, In in, Out out, T init, size_t n, hipStream_t s = 0)
{
    return rocprim::exclusive_scan(tmp, bytes, in, out, init, n, rocprim::plus<T>(), s);
}
template <typename In, typename Keys, typename Counts, typename NRuns>
static inline hipError_t tf_run_length_encode(void *tmp, size_t &bytes, In in, Keys unique_out, Counts counts_out, NRuns n_runs_out, size_t n, hipStream_t s = 0)
{
    return rocprim::run_length_encode(tmp, bytes, in, (unsigned int)n, unique_out, counts_out, n_runs_out, s);
}
template <typename K, typename V>
static inline hipError_t tf_sort_pairs(void *tmp, size_t &bytes, const K *keys_in, K *keys_out, const V *vals_in, V *vals_out, size_t n, hipStream_t s = 0)
{
    return rocprim::radix_sort_pairs(tmp, bytes, keys_in, keys_out, vals_in, vals_out, n, 0, 8 * sizeof(K), s);
}
template <typename K, typename V, typename NOut>
static inline hipError_t tf_sum_by_key(void *tmp, size_t &bytes, const K *keys_in, K *unique_out, const V *vals_in, V *sums_out, NOut n_out, size_t n, hipStream_t s = 0)
{
    return rocprim::reduce_by_key(tmp, bytes, keys_in, vals_in, (unsigned int)n, unique_out, sums_out, n_out, rocprim::plus<V>(), rocprim::equal_to<K>(), s);
}

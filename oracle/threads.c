/* threads.c — oracle-side OpenMP control (test infrastructure only).  A build without -fopenmp (`make asan`) runs on one thread. */
#define ORC_API __attribute__((visibility("default")))
#ifdef _OPENMP
#include <omp.h>
ORC_API void orc_set_num_threads(int n) { if (n > 0) omp_set_num_threads(n); }
ORC_API int orc_get_max_threads(void) { return omp_get_max_threads(); }
#else
ORC_API void orc_set_num_threads(int n) { (void)n; }
ORC_API int orc_get_max_threads(void) { return 1; }
#endif

/* threads.c — oracle-side OpenMP control (test infrastructure only). */
#include <omp.h>
#define ORC_API __attribute__((visibility("default")))
ORC_API void orc_set_num_threads(int n) { if (n > 0) omp_set_num_threads(n); }
ORC_API int orc_get_max_threads(void) { return omp_get_max_threads(); }

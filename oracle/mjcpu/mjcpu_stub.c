/* placeholder until oracle/mjcpu lands */
#include <stddef.h>
void* mjcpu_create(const char* t, int n, int s, int m, const double* e, int ne) { (void)t;(void)n;(void)s;(void)m;(void)e;(void)ne; return NULL; }
int mjcpu_num_state_keys(void* h) { (void)h; return 0; }
int mjcpu_state_key(void* h, int i, char* n, int* d, int* e) { (void)h;(void)i;(void)n;(void)d;(void)e; return -1; }
int mjcpu_action_info(void* h, int* d, int* e) { (void)h;(void)d;(void)e; return -1; }
void mjcpu_reset(void* h, const int* ids, int k, void** out) { (void)h;(void)ids;(void)k;(void)out; }
void mjcpu_step(void* h, const int* ids, int k, const void* a, void** out) { (void)h;(void)ids;(void)k;(void)a;(void)out; }
void mjcpu_destroy(void* h) { (void)h; }
int mjcpu_state_dim(void* h) { (void)h; return 0; }
void mjcpu_get_state(void* h, const int* ids, int k, double* o) { (void)h;(void)ids;(void)k;(void)o; }
void mjcpu_set_state(void* h, const int* ids, int k, const double* o) { (void)h;(void)ids;(void)k;(void)o; }

/* The C ABI from plain C (gcc -std=c99 -pedantic): build the README integrator ~(_1[_1] + _2) with the
 * expression constructors, compile it (pure host work), read the program back.  No GPU needed; on a
 * box with a GPU (argv[1] == "run") it also evaluates four samples through a 1-stream bank
 * (flowz/README.md:33-37: 1 2 3 4 -> 1 3 6 10). */
#include <stdio.h>
#include <string.h>

#include "flowz_hip.h"

int main(int argc, char** argv)
{
   fz_expr *d = fz_delayed(1, 1), *x = fz_placeholder(2), *s, *fb;
   fz_program* p = NULL;
   fz_info info;
   fz_ir_node ir[16];
   int n, i, fail = 0;
   s = fz_arith(FZ_OP_ADD, d, x);
   fb = fz_feedback(s);
   if (!d || !x || !s || !fb) { printf("constructor failed: %s\n", fz_last_error()); return 1; }
   if (fz_input_arity(fb) != 1 || fz_output_arity(fb) != 1) { printf("arity\n"); fail = 1; }
   if (fz_compile(fb, &p) != FZ_OK) { printf("compile: %s\n", fz_last_error()); return 1; }
   fz_expr_release(d); fz_expr_release(x); fz_expr_release(s); fz_expr_release(fb);
   if (fz_program_info(p, &info) != FZ_OK || info.n_in != 1 || info.n_out != 1 || info.n_state != 1 || info.n_ops != 1) {
      printf("info: in %u out %u state %u ops %u\n", info.n_in, info.n_out, info.n_state, info.n_ops);
      fail = 1;
   }
   n = fz_program_ir(p, ir, 16);
   printf("%s: %d IR nodes:", fz_version(), n);
   for (i = 0; i < n && i < 16; ++i) printf(" %u(%u,%u)", ir[i].kind, ir[i].a, ir[i].b);
   printf("\n");
   if (fz_compile(NULL, &p) == FZ_OK || strlen(fz_last_error()) == 0) { printf("null expression accepted\n"); fail = 1; }
   if (argc > 1 && strcmp(argv[1], "run") == 0) {
      fz_bank* b = NULL;
      float in[4] = {1.f, 2.f, 3.f, 4.f}, out[4] = {0.f, 0.f, 0.f, 0.f};
      if (fz_bank_create(p, 1, &b) != FZ_OK || fz_bank_process_host(b, in, out, 4) != FZ_OK) {
         printf("run: %s\n", fz_last_error());
         return 1;
      }
      printf("integrator: %g %g %g %g\n", out[0], out[1], out[2], out[3]);
      if (out[0] != 1.f || out[1] != 3.f || out[2] != 6.f || out[3] != 10.f) fail = 1;
      fz_bank_destroy(b);
   } else if (fz_device_count() == 0) {
      fz_bank* b = NULL;
      if (fz_bank_create(p, 1, &b) != FZ_E_NO_DEVICE) { printf("no device, but no FZ_E_NO_DEVICE\n"); fail = 1; }
   }
   fz_program_destroy(p);
   printf(fail ? "FAILED\n" : "ok\n");
   return fail;
}

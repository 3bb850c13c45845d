// TEST INFRASTRUCTURE.  Entry point of oracle/_ref/flye_assemble (and, with integration/flye_seam.o linked in
// front, oracle/_ref/flye_assemble_gpu): the reference's own `flye-modules assemble` stage
// (/root/reference/src/assemble/main_assemble.cpp:123-257, compiled unmodified), reached the way
// src/main.cpp:22-25 reaches it.
int assemble_main(int argc, char** argv);
int main(int argc, char** argv) { return assemble_main(argc, argv); }

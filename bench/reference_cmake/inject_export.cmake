# cmake -DIN=<reference .cc> -DOUT=<generated .cc> -DDRIVER=<name> [-DSTOKES_3D=ON] -P inject_export.cmake
# Writes a copy of a reference driver with (1) `#include "alfd_export_hook.hpp"` in front of everything and (2) one hook
# statement in front of the statement that constructs the outer FGMRES solver -- found by a regular expression on the
# identifiers, no reference text is stored here (the anchors stop short of the statement's semicolon: a match that
# contains one would count as two list elements in CMake).  Fails loudly when the anchor is not found exactly once.
file(READ "${IN}" text)
if(DRIVER STREQUAL "stokes_immersed_boundary")
  set(anchor "SolverFGMRES<BlockVector<double>>[ \t\r\n]+solver_fgmres\\(outer_solver_control\\)")
  set(hook "ALFD_EXPORT_STOKES_HOOK(\"stokes_immersed_boundary.alfd\");\n      ")
elseif(DRIVER STREQUAL "immersed_laplace")
  set(anchor "SolverFGMRES<BlockVector<double>>[ \t\r\n]+solver_fgmres\\(schur_solver_control\\)")
  set(hook "ALFD_EXPORT_LAPLACE_HOOK(\"immersed_laplace.alfd\");\n    ")
else()
  message(FATAL_ERROR "no export hook for driver ${DRIVER}")
endif()
string(REGEX MATCHALL "${anchor}" hits "${text}")
list(LENGTH hits n)
if(NOT n EQUAL 1)
  message(FATAL_ERROR "${DRIVER}: expected the FGMRES anchor exactly once, found ${n} -- has the reference changed?")
endif()
string(REGEX REPLACE "(${anchor})" "${hook}\\1" text "${text}")
if(STOKES_3D AND DRIVER STREQUAL "stokes_immersed_boundary")
  # main() carries `const unsigned int dim = 1, spacedim = 2;` with the 3-D line commented out below it
  string(REGEX MATCHALL "const unsigned int dim = 1, spacedim = 2" hits3 "${text}")
  list(LENGTH hits3 n3)
  if(NOT n3 EQUAL 1)
    message(FATAL_ERROR "stokes_immersed_boundary: dim / spacedim line not found")
  endif()
  string(REPLACE "const unsigned int dim = 1, spacedim = 2;" "const unsigned int dim = 2, spacedim = 3;" text "${text}")
endif()
file(WRITE "${OUT}" "#include \"alfd_export_hook.hpp\"\n${text}")
message(STATUS "alfd export hook inserted into a copy of ${IN} -> ${OUT}")

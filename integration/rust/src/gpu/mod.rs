//! GPU witness engine binding (libh2e.so, include/h2e.h).  UNVERIFIED source: see integration/rust/README.md.
pub mod context;
pub mod ffi;

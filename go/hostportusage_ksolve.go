//go:build cgo && ksolve

// hostportusage_ksolve.go — goes into pkg/scheduling (package scheduling, next to hostportusage.go). The ksolve flattener
// (pkg/controllers/provisioning/scheduling/ksolve_flatten.go) encodes host ports as bits over the problem's distinct
// <hostIP, hostPort, protocol> triples and needs to read which triples a node / a daemon-overhead group already holds;
// HostPortUsage keeps them in an unexported map (hostportusage.go:35-37).
//
// NOT COMPILED IN THIS REPOSITORY'S IMAGE (no Go toolchain).
package scheduling

// Reserved returns every host port currently held, in no particular order (read-only use).
func (u *HostPortUsage) Reserved() []HostPort {
	if u == nil {
		return nil
	}
	var out []HostPort
	for _, ports := range u.reserved {
		out = append(out, ports...)
	}
	return out
}
